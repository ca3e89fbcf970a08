"""Bank-conflict check (host arithmetic, MI355X_MICROARCH.md LDS table) for the fragment reads of the 256x256 bf16 core
(csrc/bgemm256_core.h): the v_mfma_f32_16x16x32_bf16 operand reads from the KC image (ds_read_b128) and the MC image
(ds_read_b64_tr_b16).  Prints the worst number of distinct addresses per bank per lane group (1 = conflict-free)."""
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
HALVES = [list(range(32)), list(range(32, 64))]


def worst(addr, groups, width):
    w = 0
    for g in groups:
        banks = {}
        for l in g:
            for b in range(addr(l) // 4, (addr(l) + width) // 4):
                banks.setdefault(b % 64, set()).add(addr(l) // 4 * 4 if width <= 4 else (addr(l), b))
        # distinct dwords per bank
        per = {}
        for l in g:
            for d in range(width // 4):
                dw = addr(l) // 4 + d
                per.setdefault(dw % 64, set()).add(dw)
        w = max(w, max(len(v) for v in per.values()))
    return w


def kc_addr(R0, t, s):
    def f(l):
        i, q = l & 15, l >> 4
        r = R0 + 16 * t + i
        return r * 128 + (((4 * s + q) ^ ((r >> 1) & 7)) << 4)
    return f


def mc_addr(C0, t, s, hf):
    def f(l):
        qg, a, b = l >> 4, (l & 15) >> 2, l & 3
        k = 32 * s + 8 * qg + 4 * hf + a
        ch = (C0 + 16 * t) // 8 + (b >> 1)
        key = ((k & 3) << 2) | ((k >> 2) & 3)
        return 256 * k + 16 * (ch ^ key) + 8 * (b & 1)
    return f


if __name__ == "__main__":
    wk = max(worst(kc_addr(R0, t, s), B128_GROUPS, 16) for R0 in (0, 32, 64, 96) for t in range(4) if R0 + 16 * t < 128 for s in range(2))
    wm = max(worst(mc_addr(C0, t, s, hf), HALVES, 8) for C0 in (0, 32, 64, 96) for t in range(4) if C0 + 16 * t < 128 for s in range(2) for hf in range(2))
    print("KC image, ds_read_b128 (16x16x32 operand): worst distinct dwords per bank per lane group =", wk)
    print("MC image, ds_read_b64_tr_b16 (16x16x32 operand): worst =", wm)
