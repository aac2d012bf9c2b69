"""World-size-2 (and 3) gloo tests of the K-sharded layer's host logic on CPU: sharding plan,
partial-sum exchange placed before the LayerNorm, both exchange modes, ragged token counts.
The compute callbacks are oracle-backed here (test infrastructure); on a GPU box the same control
flow runs with the HIP callbacks over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from onebit_amd.sharded import k_range, k_sharded_forward, shard_k


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _np_partial(shard, x_slice):
    from oracle.oracle import np_int8_to_fp
    W = np_int8_to_fp(shard.weight.numpy(), np.float64)
    xs = x_slice.numpy()
    if xs.dtype == np.float16:
        a = (xs.astype(np.float32) * shard.input_factor.numpy().astype(np.float32)[None]).astype(np.float16)
    else:
        a = xs * shard.input_factor.numpy()[None]
    return torch.from_numpy((a.astype(np.float64) @ W.T).astype(np.float32))


def _np_epilogue(shard, z, dtype, eps):
    zf = z.numpy()
    g = shard.weight_scale.numpy()
    if dtype == torch.float16:
        u = (zf.astype(np.float16).astype(np.float32) * g.astype(np.float32)[None]).astype(np.float16).astype(np.float64)
    else:
        u = (zf * g[None]).astype(np.float64)
    mean = u.mean(-1, keepdims=True)
    var = ((u - mean) ** 2).mean(-1, keepdims=True)
    y = (u - mean) / np.sqrt(var + eps)
    if shard.bias is not None:
        y = y.astype(np.float16 if dtype == torch.float16 else np.float32).astype(np.float64) + shard.bias.numpy()[None]
    return torch.from_numpy(y.astype(np.float16 if dtype == torch.float16 else np.float32))


def _worker(rank, world, port, T, K, N, mode, dtype_name, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dt = torch.float16 if dtype_name == "f16" else torch.float32
        g = torch.Generator().manual_seed(5)                      # identical full tensors on every rank
        W = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8)
        h = (0.1 * (0.5 + torch.rand(K, generator=g))).to(dt)
        gs = (0.1 * (0.5 + torch.rand(N, generator=g))).to(dt)
        b = (0.1 * torch.randn(N, generator=g)).to(dt)
        x = torch.randn(T, K, generator=g).to(dt)
        shard = shard_k(W, h, gs, b, rank, world, copy=(rank % 2 == 0))   # exercise view and copy
        assert shard.k0 % 32 == 0 and shard.k1 % 32 == 0
        y = k_sharded_forward(shard, x, mode=mode, partial_fn=_np_partial, epilogue_fn=_np_epilogue)
        assert y.shape == (T, N) and y.dtype == dt
        # every rank ends with the same complete result
        ys = [torch.empty_like(y) for _ in range(world)]
        dist.all_gather(ys, y)
        for other in ys:
            assert torch.equal(other, y)
        if rank == 0:
            from oracle.oracle import COracle
            c = COracle()
            fn = c.forward_f16 if dt == torch.float16 else c.forward_f32
            ref = fn(W.numpy(), x.numpy(), h.numpy(), gs.numpy(), b.numpy())
            err = np.abs(y.numpy().astype(np.float32) - ref.astype(np.float32)).max()
            out.put(float(err))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T,K,N,mode,dtype", [
    (2, 5, 256, 48, "rs_ag", "f16"),          # T not divisible by world: padded reduce_scatter
    (2, 4, 256, 48, "allreduce", "f16"),
    (3, 7, 1376 // 32 * 32, 40, "rs_ag", "f32"),   # uneven K split (43 dwords over 3 ranks)
])
def test_k_sharded_forward_gloo(world, T, K, N, mode, dtype):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, K, N, mode, dtype, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    err = q.get(timeout=5)
    assert err <= (4e-3 if dtype == "f16" else 2e-4), err


def test_k_range_partition():
    for K in (4096, 11008, 5120, 13824, 1376, 32):
        for world in (1, 2, 3, 4, 8):
            if K // 32 < world:
                continue
            edges = [k_range(K, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == K
            for (a0, a1), (b0, b1) in zip(edges, edges[1:]):
                assert a1 == b0 and a0 < a1
            assert all(a % 32 == 0 and b % 32 == 0 for a, b in edges)
    with pytest.raises(ValueError):
        k_range(40, 0, 2)


def _model_worker(rank, world, port, golden_dir, mode, out):
    """Whole tiny model, every 1-bit layer K-sharded (BASELINE config 4 at toy size): prefill + 4
    greedy decode steps against the logits recorded from the reference model."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
        from onebit_amd.sharded import KShardedBitLinear, shard_model_k
        z = np.load(os.path.join(golden_dir, "model_tiny_b.npz"))
        kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
        model = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float32)
        model.load_state_dict({k[3:]: (torch.from_numpy(z[k]) if z[k].dtype == np.int8 else torch.from_numpy(z[k]).float())
                               for k in z.files if k.startswith("sd_")})
        shard_model_k(model.eval(), rank, world, mode=mode, partial_fn=_np_partial, epilogue_fn=_np_epilogue)
        n = sum(isinstance(m, KShardedBitLinear) for m in model.modules())
        assert n == 7 * kw["num_hidden_layers"]
        ids = torch.from_numpy(z["input_ids"])
        cache = model.new_cache(1, ids.shape[1] + 8)
        logits = model(ids, cache)
        errs = [float(np.abs(logits[0].numpy() - z["prefill_logits_f32"][0]).max())]
        toks = z["greedy_f32"][0]
        for i in range(4):
            lg = model(torch.tensor([[int(toks[i])]]), cache)
            errs.append(float(np.abs(lg[0, -1].numpy() - z["decode_logits_f32"][0][i]).max()))
        gathered = [None] * world
        dist.all_gather_object(gathered, errs)
        assert all(g == errs for g in gathered)                   # every rank computed the same logits
        if rank == 0:
            out.put((max(errs), float(np.abs(z["prefill_logits_f32"]).max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "allreduce"), (2, "rs_ag")])
def test_k_sharded_model_decode_gloo(golden_dir, world, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, world, port, golden_dir, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    err, scale = q.get(timeout=5)
    assert err <= 2e-3 * max(1.0, scale), (err, scale)


def _static_worker(rank, world, port, golden_dir, out):
    """StaticShapeDecoder (device-side token / position, static shapes: the graph-capturable config-4 step) on the
    K-sharded tiny model: greedy tokens and logits against the reference's recorded decode."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
        from onebit_amd.sharded import StaticShapeDecoder, shard_model_k
        z = np.load(os.path.join(golden_dir, "model_tiny_b.npz"))
        kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
        model = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float32)
        model.load_state_dict({k[3:]: (torch.from_numpy(z[k]) if z[k].dtype == np.int8 else torch.from_numpy(z[k]).float())
                               for k in z.files if k.startswith("sd_")})
        shard_model_k(model.eval(), rank, world, mode="allreduce", partial_fn=_np_partial, epilogue_fn=_np_epilogue)
        ids = torch.from_numpy(z["input_ids"])
        dec = StaticShapeDecoder(model, max_len=ids.shape[1] + 8, use_graph=False)
        toks = z["greedy_f32"][0]
        first = dec.prime(ids)
        assert first == int(toks[0])
        errs = []
        for i in range(4):
            dec.step()
            errs.append(float(np.abs(dec._logits[0, -1].numpy() - z["decode_logits_f32"][0][i]).max()))
            assert int(dec.tok.item()) == int(toks[i + 1]) and int(dec.pos.item()) == ids.shape[1] + i + 1
        assert dec.out_tokens[ids.shape[1]:ids.shape[1] + 4].tolist() == [int(t) for t in toks[1:5]]
        gathered = [None] * world
        dist.all_gather_object(gathered, errs)
        assert all(g == errs for g in gathered)
        if rank == 0:
            out.put((max(errs), float(np.abs(z["decode_logits_f32"]).max())))
    finally:
        dist.destroy_process_group()


def test_static_shape_decoder_k_sharded_gloo(golden_dir):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_static_worker, args=(r, world, port, golden_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    err, scale = q.get(timeout=5)
    assert err <= 2e-3 * max(1.0, scale), (err, scale)


def _np_rows_u(shard, x):
    """Oracle stand-in for hip_rows_u: pre-LayerNorm u of the rank's rows."""
    from oracle.oracle import np_forward_f16, np_forward_f32
    fn = np_forward_f16 if x.dtype == torch.float16 else np_forward_f32
    _, u = fn(shard.weight.numpy(), x.numpy(), shard.input_factor.numpy(), shard.weight_scale.numpy(), None,
              return_pre_ln=True)
    return torch.from_numpy(np.ascontiguousarray(u))


def _n_worker(rank, world, port, T, K, N, dtype_name, gather, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from onebit_amd.sharded import n_sharded_forward, shard_n
        dt = torch.float16 if dtype_name == "f16" else torch.float32
        g = torch.Generator().manual_seed(11)
        W = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).view(torch.int8)
        h = (0.1 * (0.5 + torch.rand(K, generator=g))).to(dt)
        gs = (0.1 * (0.5 + torch.rand(N, generator=g))).to(dt)
        b = (0.1 * torch.randn(N, generator=g)).to(dt)
        x = torch.randn(T, K, generator=g).to(dt)
        shard = shard_n(W, h, gs, b, rank, world)
        assert shard.n0 % 16 == 0 and shard.weight.shape[0] == shard.n1 - shard.n0
        from onebit_amd.sharded import _torch_normalize, _torch_row_stats
        y = n_sharded_forward(shard, x, gather=gather, rows_fn=_np_rows_u, stats_fn=_torch_row_stats, normalize_fn=_torch_normalize)
        from oracle.oracle import COracle
        c = COracle()
        fn = c.forward_f16 if dt == torch.float16 else c.forward_f32
        ref = fn(W.numpy(), x.numpy(), h.numpy(), gs.numpy(), b.numpy()).astype(np.float32)
        if not gather:
            ref = ref[:, shard.n0:shard.n1]
        assert y.shape == ref.shape and y.dtype == dt
        errs = [None] * world
        dist.all_gather_object(errs, float(np.abs(y.numpy().astype(np.float32) - ref).max()))
        if rank == 0:
            out.put(max(errs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T,K,N,dtype,gather", [
    (2, 5, 256, 64, "f16", True),
    (2, 3, 256, 48, "f32", False),        # uneven rows: 32 + 16
    (3, 4, 128, 96, "f16", False),
])
def test_n_sharded_forward_gloo(world, T, K, N, dtype, gather):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_n_worker, args=(r, world, port, T, K, N, dtype, gather, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    err = q.get(timeout=5)
    assert err <= (4e-3 if dtype == "f16" else 2e-4), err


def test_n_range_partition():
    from onebit_amd.sharded import n_range
    for N, world in ((11008, 8), (4096, 4), (5120, 3), (40, 3)):
        cuts = [n_range(N, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == N
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        assert all(c[0] % 16 == 0 for c in cuts)


# ---------------------------------------------------------------------------------------------------
# FusedKShardedDecoder (config 4, round 5): the exchange protocol of onebit_decode_step_ksharded -- which fp32 buffer is
# all-reduced between which segments, which columns a rank multiplies, 4 collectives per layer -- under gloo with a plain
# torch statement of the five segments standing in for the C ABI (the product backend has no CPU form).
# ---------------------------------------------------------------------------------------------------
def _torch_segments_factory(layers_full, embed, final_w, lm_head):
    """layers_full[l][name] = (signs [N, K] fp32 (+1 / -1), h [K], g [N], bias or None); semantics: bitnet.py:112-122 with the
    K sum split before :115, modeling_bitllama.py:900-918 around it."""
    import math
    from onebit_amd.llama import _rotate_half

    class TorchSegments:
        def __init__(self, dec):
            self.d = dec
            self.cos, self.sin = dec.model._rope_tables(dec.dev, dec.buf["x"].dtype, dec.max_len)

        @staticmethod
        def _ln(u, bias=None, eps=1e-5):
            y = torch.nn.functional.layer_norm(u, u.shape[-1:], eps=eps)
            return y if bias is None else y + bias

        def _rms(self, h, w):
            v = h.float().pow(2).mean(-1, keepdim=True)
            return w * (h.float() * torch.rsqrt(v + self.d.cfg.rms_norm_eps)).to(h.dtype)

        def _partial(self, name, l, x):
            W, h, _, _ = layers_full[l][name]
            k0, k1 = self.d.kr(W.shape[1])
            return (W[:, k0:k1] @ (x[k0:k1] * h[k0:k1]).float()).float()

        def segment(self, l, seg):
            d, cfg, b = self.d, self.d.cfg, self.d.buf
            H, I, D = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
            nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
            NQ, NK = nh * D, nkv * D
            hA, hB = b["hres0"], b["hres1"]
            lay = d.model.model.layers
            if seg == 0:
                if l == 0:
                    hB.copy_(embed[int(d.token.item())])
                else:
                    _, _, g, _ = layers_full[l - 1]["down"]
                    hB.copy_(hA + self._ln((d.z_down * g).to(hA.dtype)))
                b["x"].copy_(self._rms(hB, lay[l].input_layernorm.weight))
                d.z_qkv.copy_(torch.cat([self._partial(n, l, b["x"]) for n in ("q", "k", "v")]))
            elif seg == 1:
                pos = int(d.pos.item())
                us = []
                for n, lo, hi in (("q", 0, NQ), ("k", NQ, NQ + NK), ("v", NQ + NK, NQ + 2 * NK)):
                    _, _, g, bias = layers_full[l][n]
                    us.append(self._ln((d.z_qkv[lo:hi] * g).to(hA.dtype), bias))
                q, k, v = us[0].view(nh, 1, D), us[1].view(nkv, 1, D), us[2].view(nkv, 1, D)
                c, s_ = self.cos[pos][None, None], self.sin[pos][None, None]
                q = q * c + _rotate_half(q) * s_
                k = k * c + _rotate_half(k) * s_
                kc, vc = d.cache.layers[l]
                kc[0, :, pos] = k[:, 0]
                vc[0, :, pos] = v[:, 0]
                keys, vals = kc[0, :, :pos + 1], vc[0, :, :pos + 1]
                if nkv != nh:
                    keys, vals = keys.repeat_interleave(nh // nkv, 0), vals.repeat_interleave(nh // nkv, 0)
                w = torch.matmul(q, keys.transpose(1, 2)) / math.sqrt(D)
                w = torch.softmax(w.float(), -1).to(q.dtype)
                b["attn_out"].copy_(torch.matmul(w, vals).reshape(NQ))
                d.z_o.copy_(self._partial("o", l, b["attn_out"]))
            elif seg == 2:
                _, _, g, bias = layers_full[l]["o"]
                hA.copy_(hB + self._ln((d.z_o * g).to(hA.dtype), bias))
                b["x"].copy_(self._rms(hA, lay[l].post_attention_layernorm.weight))
                d.z_gu.copy_(torch.cat([self._partial(n, l, b["x"]) for n in ("gate", "up")]))
            elif seg == 3:
                gg, gu = layers_full[l]["gate"][2], layers_full[l]["up"][2]
                act = torch.nn.functional.silu(self._ln((d.z_gu[:I] * gg).to(hA.dtype))) * self._ln((d.z_gu[I:] * gu).to(hA.dtype))
                b["act"].copy_(act)
                d.z_down.copy_(self._partial("down", l, b["act"]))
            else:
                _, _, g, _ = layers_full[cfg.num_hidden_layers - 1]["down"]
                hB.copy_(hA + self._ln((d.z_down * g).to(hA.dtype)))
                logits = (self._rms(hB, final_w) @ lm_head.t()).float()
                b["logits"].copy_(logits.to(b["logits"].dtype))
                nxt = int(logits.argmax())
                pos = int(d.pos.item())
                d.out_tokens[pos] = nxt
                d.token.fill_(nxt)
                d.pos.add_(1)
    return TorchSegments


def _tiny_full_model(golden_dir):
    from onebit_amd.llama import OneBitLlamaConfig, OneBitLlamaForCausalLM
    z = np.load(os.path.join(golden_dir, "model_tiny_b.npz"))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    model = OneBitLlamaForCausalLM(OneBitLlamaConfig(**kw), torch.float32)
    model.load_state_dict({k[3:]: (torch.from_numpy(z[k]) if z[k].dtype == np.int8 else torch.from_numpy(z[k]).float())
                           for k in z.files if k.startswith("sd_")})
    model.eval()
    from oracle.oracle import np_int8_to_fp
    full = []
    for layer in model.model.layers:
        a, m = layer.self_attn, layer.mlp
        ent = {}
        for n, p in (("q", a.q_proj), ("k", a.k_proj), ("v", a.v_proj), ("o", a.o_proj), ("gate", m.gate_proj), ("up", m.up_proj), ("down", m.down_proj)):
            ent[n] = (torch.from_numpy(np_int8_to_fp(p.weight.data.numpy(), np.float32)), p.input_factor.data.clone(), p.weight_scale.data.clone(),
                      None if p.bias is None else p.bias.data.clone())
        full.append(ent)
    factory = _torch_segments_factory(full, model.model.embed_tokens.weight.data, model.model.norm.weight.data, model.lm_head.weight.data)
    return z, model, factory


def _check_fused_decode(dec, z, ids, step_fn):
    toks = z["greedy_f32"][0]
    errs = []
    for i in range(4):
        step_fn()
        errs.append(float(np.abs(dec.logits().numpy() - z["decode_logits_f32"][0][i]).max()))
        assert int(dec.token.item()) == int(toks[i + 1]) and int(dec.pos.item()) == ids.shape[1] + i + 1
    assert dec.out_tokens[ids.shape[1]:ids.shape[1] + 4].tolist() == [int(t) for t in toks[1:5]]
    return errs


def _fused_worker(rank, world, port, golden_dir, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from onebit_amd.sharded import FusedKShardedDecoder, shard_model_k
        z, model, factory = _tiny_full_model(golden_dir)
        # (the prompt goes through the module path, which on CPU needs the oracle-backed callbacks of the tests above)
        shard_model_k(model, rank, world, mode="allreduce", partial_fn=_np_partial, epilogue_fn=_np_epilogue)
        ids = torch.from_numpy(z["input_ids"])
        dec = FusedKShardedDecoder(model, rank, world, max_len=ids.shape[1] + 8, use_graph=False, backend=factory, granule=32)
        assert dec.collectives_per_token == 4 * model.config.num_hidden_layers
        calls = []
        real = dist.all_reduce
        def counting(t, *a, **k):
            calls.append(tuple(t.shape))
            return real(t, *a, **k)
        assert dec.prime(ids) == int(z["greedy_f32"][0][0])
        dist.all_reduce = counting
        try:
            errs = _check_fused_decode(dec, z, ids, dec.step)
        finally:
            dist.all_reduce = real
        cfg = model.config
        NQ = cfg.num_attention_heads * cfg.head_dim
        per_layer = [(NQ + 2 * cfg.num_key_value_heads * cfg.head_dim,), (cfg.hidden_size,), (2 * cfg.intermediate_size,), (cfg.hidden_size,)]
        assert calls == per_layer * cfg.num_hidden_layers * 4          # 4 collectives per layer and token, q|k|v and gate|up merged
        gathered = [None] * world
        dist.all_gather_object(gathered, errs)
        assert all(g == errs for g in gathered)                         # every rank decoded the same logits
        if rank == 0:
            out.put((max(errs), float(np.abs(z["decode_logits_f32"]).max())))
    finally:
        dist.destroy_process_group()


def test_fused_k_sharded_decoder_gloo(golden_dir):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fused_worker, args=(r, world, port, golden_dir, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    err, scale = q.get(timeout=5)
    assert err <= 2e-3 * max(1.0, scale), (err, scale)


@pytest.mark.parametrize("world", [1, 2, 4])
def test_fused_k_sharded_decoder_lockstep_ranks(golden_dir, world):
    """The same protocol with all ranks in ONE process (lockstep_step: what the GPU test does on one device): slices of
    1 / 2 / 4 ranks (K = 128 and 352 over 4 ranks: ragged granule counts), every rank ends with the reference's tokens."""
    from onebit_amd.sharded import FusedKShardedDecoder, lockstep_step
    z, model, factory = _tiny_full_model(golden_dir)
    ids = torch.from_numpy(z["input_ids"])
    # prompt through a CPU-capable copy of the model (world-1 shard with the oracle-backed callbacks), cache copied to every rank
    from onebit_amd.sharded import shard_model_k
    shard_model_k(model, 0, 1, mode="allreduce", partial_fn=_np_partial, epilogue_fn=_np_epilogue)
    decs = [FusedKShardedDecoder(model, r, world, max_len=ids.shape[1] + 8, use_graph=False, backend=factory, granule=32,
                                 reduce_fn=lambda t: None) for r in range(world)]
    first = [d.prime(ids) for d in decs]
    assert all(f == int(z["greedy_f32"][0][0]) for f in first)
    errs = _check_fused_decode(decs[0], z, ids, lambda: lockstep_step(decs))
    for d in decs[1:]:
        assert torch.equal(d.out_tokens, decs[0].out_tokens) and torch.equal(d.buf["logits"], decs[0].buf["logits"])
    assert max(errs) <= 2e-3 * max(1.0, float(np.abs(z["decode_logits_f32"]).max()))
    edges = [d.kr(352) for d in decs]
    assert edges[0][0] == 0 and edges[-1][1] == 352 and all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
