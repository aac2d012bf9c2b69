"""Rank process of test_fused_k_sharded_decoder_two_ranks_rccl: `world` ranks, one GPU each, RCCL all-reduces -- first eager, then captured
with the kernels in ONE HIP graph (capture agreed across ranks) -- against the lockstep emulation of the same ranks on rank 0's device."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    from onebit_amd.llama import OneBitLlamaConfig, build_synthetic_model
    from onebit_amd.sharded import FusedKShardedDecoder, lockstep_step
    cfg = OneBitLlamaConfig(vocab_size=640, hidden_size=1024, intermediate_size=2816, num_hidden_layers=3, num_attention_heads=8,
                            max_position_embeddings=64)
    model = build_synthetic_model(cfg, seed=31, device=dev)                      # the same checkpoint on every rank
    ids = torch.randint(0, cfg.vocab_size, (1, 7), generator=torch.Generator().manual_seed(3)).to(dev)
    # reference: all ranks emulated in lockstep on this device (what tests/test_gpu_config4.py holds to the engine and the goldens)
    emu = [FusedKShardedDecoder(model, r, world, max_len=32, use_graph=False, reduce_fn=lambda t: None) for r in range(world)]
    for d in emu:
        d.prime(ids)
    ref = []
    for _ in range(6):
        lockstep_step(emu)
        torch.cuda.synchronize(dev)
        ref.append(emu[0].buf["logits"].clone())
    for use_graph in (False, True):
        dec = FusedKShardedDecoder(model, rank, world, max_len=32, use_graph=False)
        dec.prime(ids)
        if use_graph:
            dec.use_graph = True
            ok = torch.tensor([1 if dec.capture() else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            assert int(ok.item()) == 1, "graph capture with RCCL all-reduces failed on some rank"
        for i in range(6):
            dec.step()
            torch.cuda.synchronize(dev)
            got = dec.buf["logits"]
            # the ring all-reduce adds the ranks' partial sums in another order than the emulation's stack().sum(): fp32 round-off
            err = float((got.float() - ref[i].float()).abs().max())
            assert err <= 4e-3 * float(ref[i].float().abs().max()), (use_graph, i, err)
            assert int(got.float().argmax()) == int(ref[i].float().argmax()) or err > 0
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("kshard-rccl ok")


if __name__ == "__main__":
    main()
