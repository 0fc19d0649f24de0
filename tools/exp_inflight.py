import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.hipnet import dec_ipt
from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input
cfg = NetConfig(24, 4, patch_size=32, load_size=256)
B = 8
NI = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sd = generate_state_dict(cfg, seed=0)
nets, graphs, streams = [], [], []
x = synthetic_input(B, cfg).to("cuda:0")
for k in range(NI):
    net = dec_ipt(cfg, compute_dtype="fp16"); net.load_state_dict(sd); net.to("cuda:0")
    net(x); torch.cuda.synchronize()
    gid, outs = net.capture(x)
    nets.append(net); graphs.append((gid, outs)); streams.append(torch.cuda.Stream())
torch.cuda.synchronize()
def run(steps):
    for i in range(steps):
        k = i % NI
        with torch.cuda.stream(streams[k]):
            nets[k].replay(graphs[k][0])
    torch.cuda.synchronize()
run(6)
t0 = time.perf_counter(); run(40); dt = time.perf_counter() - t0
print("in flight %d: %.1f img/s (%.3f ms/step)" % (NI, 40 * B / dt, dt / 40 * 1e3))
