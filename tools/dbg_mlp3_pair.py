"""k_mlp3p (knob mlp3.pair) against the shipped k_mlp3<24, ...>: one launch alone, timing + error against an fp64 restatement + stamps.
  python3 tools/dbg_mlp3_pair.py [variants, e.g. 9] [stamped variant, e.g. 10]   (profiles/r06_mlp3_pair_variants.txt)"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from cfen_vit_dehazing_amd import ops, packing
d = "cuda:0"
flush = torch.empty(256 << 20, dtype=torch.uint8, device=d)
def timeit(f, n=12, cold=True):
    f(); f(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        if cold:
            flush.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
D, H = 384, 1536
torch.manual_seed(0)
VARIANTS = [int(v) for v in sys.argv[1].split(",") if v] if len(sys.argv) > 1 else []
STAMPED = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for M in (1000, 2048, 24576):
    r = lambda *s, sc=1.0: (torch.randn(*s, device=d) * sc).half()
    x, att = r(M, D), r(M, D)
    wp, w1a, w2a, w1b, w2b = r(D, D, sc=D ** -0.5), r(H, D, sc=D ** -0.5), r(D, H, sc=H ** -0.5), r(H, D, sc=D ** -0.5), r(D, H, sc=H ** -0.5)
    g, b = 1 + 0.1 * torch.randn(D, device=d), 0.1 * torch.randn(D, device=d)
    b1a, b2a, b1b, b2b = (torch.randn(n, device=d) * 0.1 for n in (H, D, H, D))
    kd, kh = packing.kperm32(D).to(d), packing.kperm32(H).to(d)
    sa, sb, sp = packing.pack_stream_pair(w1a[:, kd], w2a[:, kh]), packing.pack_stream_pair(w1b[:, kd], w2b[:, kh]), packing.pack_stream_sq(wp)
    call = lambda: ops.mlp_stream_block(x, sa, b1a, b2a, H, ln=(g, b), second=(sb, b1b, b2b), proj=(att, sp))
    ops.tune("mlp3.pair", 0)
    ref = call().float()
    # fp64 reference
    X = x.double() + att.double() @ wp.double().t()
    mu = X.mean(1, keepdim=True); var = ((X - mu) ** 2).mean(1, keepdim=True)
    Ln = (X - mu) / torch.sqrt(var + 1e-5) * g.double() + b.double()
    Y1 = X + torch.relu(Ln.half().double() @ w1a.double().t() + b1a.double()).half().double() @ w2a.double().t() + b2a.double()
    Y2 = Y1 + torch.relu(Y1.half().double() @ w1b.double().t() + b1b.double()).half().double() @ w2b.double().t() + b2b.double()
    t0 = timeit(call)
    ops.tune("mlp3.pair", 1)
    out = call().float()
    torch.cuda.synchronize()
    again = call().float()
    t1 = timeit(call)
    print("   warm caches (no 256 MiB fill between launches): pair %.1f us" % timeit(call, cold=False), flush=True)
    for v in VARIANTS:
        ops.tune("mlp3.pair", v)
        o2 = call().float()
        torch.cuda.synchronize()
        o3 = call().float()
        print("   variant %d: %.1f us (max |. - pair| %.3e, vs fp64 %.3e, deterministic %s)" % (v, timeit(call), float((o2 - out).abs().max()), float((o2.double() - Y2).abs().max()), bool(torch.equal(o2, o3))), flush=True)
    sys.stderr.write("==== M=%d\n" % M); sys.stderr.flush()
    ops.tune("mlp3.pair", STAMPED)
    for _ in range(2):
        flush.fill_(1); call(); torch.cuda.synchronize()
    ops.tune("mlp3.pair", 0)
    e_ref = float((ref.double() - Y2).abs().max()); e_pair = float((out.double() - Y2).abs().max())
    print("M=%d: shipped %.1f us, pair %.1f us; max|pair - shipped| %.3e; vs fp64: shipped %.3e pair %.3e; pair deterministic %s" % (M, t0, t1, float((out - ref).abs().max()), e_ref, e_pair, bool(torch.equal(out, again))), flush=True)
