#!/bin/bash
# Diagnostic build with per-phase s_memtime stamps (NOT the product build): builds a private copy of the
# library under /tmp and prints the mean cycles per K-tile phase for a few conv shapes.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
D=/tmp/dcap_stamps
rm -rf $D && mkdir -p $D/image-captioning_amd $D/include
cp -r $ROOT/image-captioning_amd/*.py $D/image-captioning_amd/
cp -r $ROOT/image-captioning_amd/csrc $D/image-captioning_amd/csrc
cp -r $ROOT/image_captioning_amd $D/
cp $ROOT/include/dcap.h $D/include/
cd $D/image-captioning_amd/csrc
for f in gemm gemm_tn conv conv_bs lstm loss roialign proposal; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fgpu-rdc -DDCAP_STAMPS -c $f.hip -o $f.o 2>/dev/null &
done
wait
hipcc --offload-arch=gfx950 -fgpu-rdc -shared -fPIC -o libdcap_hip.so gemm.o gemm_tn.o conv.o conv_bs.o lstm.o loss.o roialign.o proposal.o
cd $D
python - "$@" <<'PY'
import ctypes as C, sys, torch, numpy as np
sys.path.insert(0, '/tmp/dcap_stamps')
from image_captioning_amd import ops, _lib
lib = _lib.load()
lib.dc_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
shapes = [("res4_2b 3x3 256 b1", 1, 64, 256, 256, 3), ("res4_2b 3x3 256 b2", 2, 64, 256, 256, 3), ("res4_2b b4", 4, 64, 256, 256, 3),
          ("fpn_p3 3x3 b2", 2, 128, 256, 256, 3), ("res4_2a 1x1 1024>256 b2", 2, 64, 1024, 256, 1),
          ("res4_2c 1x1 256>1024 b2", 2, 64, 256, 1024, 1), ("res3_2c 1x1 128>512 b2", 2, 128, 128, 512, 1)]
for name, B, H, Cin, Cout, k in shapes:
    x = torch.randn(B, H, H, Cin, device='cuda'); w = torch.randn(Cout, k*k*Cin, device='cuda') * 0.02
    y = torch.empty(B, H, H, Cout, device='cuda')
    run = lambda: ops.conv2d(x, w, k, k, 1, (k-1)//2, (k-1)//2, H, H, None, None, None, 0, False, out=y)
    run(); lib.dc_debug_stamps(None, 1)
    run()
    out = (C.c_ulonglong * 6)(); lib.dc_debug_stamps(out, 1)
    n = out[5]
    names = ["issue global loads", "ds_read + MFMAs", "vmcnt(0) wait", "ds_write", "barrier"]
    print(name, " wave-ktiles:", n)
    tot = sum(out[i] for i in range(5))
    for i in range(5):
        print("   %-20s %8.1f cycles/k-tile (%4.1f%%)" % (names[i], out[i] / n, 100.0 * out[i] / tot))
    print("   total %.1f" % (tot / n))
PY
