"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0 (placeholder until the model lands)."""


def run():
    import torch
    from . import ops
    a = torch.randn(64, 64, device="cuda")
    b = torch.randn(64, 64, device="cuda")
    c = ops.gemm(a, b)
    torch.cuda.synchronize()
    ref = (a.double() @ b.double()).float()
    assert float((c - ref).abs().max()) < 1e-3
