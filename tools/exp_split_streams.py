"""Experiment: the eval forward + PIT of the headline batch as TWO half batches on two streams inside one captured graph
(kernels of the two halves overlap each other's tails) against the one-stream step of bench.py.
python tools/exp_split_streams.py [nsplit]"""
import copy, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from css_amd import ops
from css_amd.models import MODELS
from css_amd.utils import synthetic as syn

NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, T, N_SAMPLES = 32, 400, 256 * 399 + 512 - 512 + 256 * 0
dev = torch.device("cuda", 0)
conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=16, kernel_size=33, dropout_rate=0.1, relative_pos_emb=True)
model = MODELS["Conformer"](257, 257, 2, 1, conf)
model.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_blocks=16), 0))
model = model.to(dev).eval()
models = [model] + [copy.deepcopy(model) for _ in range(NS - 1)]
import bench
mix, src = syn.synthetic_sources(B, bench.N_SAMPLES, seed=0)
audio = torch.stack([mix, src[0], src[1]], 0).to(dev)
streams = [torch.cuda.Stream() for _ in range(NS)]
hb = B // NS

def step_one():
    with torch.no_grad():
        mags = ops.stft(audio, ("mag",))["mag"]
        return model.forward_loss(mags[0], [mags[1], mags[2]])[0]

def step_split():
    with torch.no_grad():
        mags = ops.stft(audio, ("mag",))["mag"]
        main = torch.cuda.current_stream()
        losses = []
        for i, (m, st) in enumerate(zip(models, streams)):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                sl = slice(i * hb, (i + 1) * hb)
                losses.append(m.forward_loss(mags[0][sl], [mags[1][sl], mags[2][sl]])[0])
        for st in streams:
            main.wait_stream(st)
        return torch.stack([l.reshape(()) for l in losses]).mean()

def graphed(fn):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        out = fn()
    return g, out

def time_graph(g, n=50):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

g1, l1 = graphed(step_one)
g2, l2 = graphed(step_split)
for r in range(3):
    t1, t2 = time_graph(g1), time_graph(g2)
    print(f"one stream {t1:.4f} ms (loss {float(l1):.6f})   {NS} half-batch streams {t2:.4f} ms (loss {float(l2):.6f})")
