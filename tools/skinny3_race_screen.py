"""Race screen for the LDS-DMA skinny GEMM (ob_skinny3.h: wave-private rings, counted s_waitcnt vmcnt, DMA writes the
compiler does not see): the same pre-scaled call repeated many times on several shapes and token counts must give
bit-identical outputs every time, with competing memory traffic between repetitions and a second stream hammering the
memory system during them -- a transfer consumed before it landed, or overwritten before it was read, shows up as rare
differing outputs.  Also the whole 32-slot batched step (all seven projections + row kernels) repeated from the same state.
Usage: python tools/skinny3_race_screen.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onebit_amd import BitLinearInf
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(11)
shapes = [(32, 4096, 4096), (32, 4096, 11008), (32, 11008, 4096), (32, 5120, 13824), (16, 4096, 11008), (7, 4096, 4096),
          (64, 4096, 11008), (33, 1024, 528), (2, 640, 1000), (32, 512, 48)]
side = torch.cuda.Stream()
noise_src = torch.empty(256 << 20, dtype=torch.uint8, device=dev).random_()
noise_dst = torch.empty_like(noise_src)
bad = 0
for (T, K, N) in shapes:
    m = BitLinearInf(K, N, dtype=torch.float16).to(dev)
    m.weight.data = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8).to(dev)
    m.input_factor.data = (0.1 * (0.5 + torch.rand(K, generator=g))).half().to(dev)
    m.weight_scale.data = (0.1 * (0.5 + torch.rand(N, generator=g))).half().to(dev)
    assert m.prescaled_ok(T)
    a = (torch.randn(T, K, generator=g).half().to(dev) * m.input_factor.data)
    ref = m.pre_layernorm_prescaled(a).clone()
    torch.cuda.synchronize()
    n, diff = int(os.environ.get("REPS", "200")), 0
    for i in range(n):
        if i % 4 == 0:
            with torch.cuda.stream(side):
                noise_dst.copy_(noise_src)              # concurrent HBM traffic on another stream
        if i % 7 == 0:
            junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev).random_()
        y = m.pre_layernorm_prescaled(a)
        if not torch.equal(y, ref):
            diff += 1
    torch.cuda.synchronize()
    print("T=%3d K=%5d N=%5d: %d / %d repetitions differ" % (T, K, N, diff, n), flush=True)
    bad += diff
    del m, a, ref

# the batched step: same tokens / positions / caches every time -> same logits every time
from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
from onebit_amd.engine import BatchedDecodeStep
cfg = OneBitLlamaConfig.llama_7b()
cfg.num_hidden_layers = 4
model = build_synthetic_model(cfg, seed=2, device=dev)
B, max_len = 32, 32
cache = model.new_cache(B, max_len)
ids = torch.randint(0, cfg.vocab_size, (B, 6), generator=g).to(dev)
with torch.no_grad():
    model(ids, cache)
step = BatchedDecodeStep(model, cache.layers, B, max_len, sample=True, keep_logits=True)
step.tokens.copy_(torch.randint(0, cfg.vocab_size, (B,), generator=g).to(torch.int32))
step.pos.fill_(6)
step.launch(); torch.cuda.synchronize()
ref = step.logits.clone()
diff = 0
for i in range(100):
    if i % 4 == 0:
        with torch.cuda.stream(side):
            noise_dst.copy_(noise_src)
    step.launch()                                       # position 6 again: the cache row is rewritten with the same values
    torch.cuda.synchronize()
    if not torch.equal(step.logits, ref):
        diff += 1
print("batched step, 32 slots, 4 layers of 7B width: %d / 100 repetitions differ" % diff)
bad += diff
print("race screen:", "CLEAN" if bad == 0 else "FAILED")
sys.exit(1 if bad else 0)
