// EXPERIMENT (round 4), not part of the product: the block body of ob_flash.h's kernel with the two query tiles of a wave taken one
// after the other, the exp2 / row-sum / convert stream of one tile hand-interleaved with the other tile's MFMAs (asm statements,
// so that the order survives the optimiser):
//   A  scores(q0)                    B  scores(q1) + exponentials(q0)
//   C  output(q0) + exponentials(q1) D  output(q1)
// The generated code is exactly that (per step: MFMA, fragment read 4 steps ahead, 3-4 VALU), correct at every shape flash_lab
// checks -- and SLOWER: 577 TFLOP/s against 645 for the plain body at 8 x 2048 x 32 x 128.  K and V fragments are read once per tile
// (96 LDS reads per block instead of 48), and with two such waves on a SIMD its issue port is the limit (tools/pipe_overlap_probe.hip,
// mv2: two interleaved MFMA + VALU waves take twice as long as one; an MFMA costs the port about 8 cycles, a VALU op 2.75).
// To try it again: paste `helpers` before the kernel template and `body` before `const int nfull` in ob_flash_fwd_kernel, and call
// block_main(kb) for kb < nfull.
#if 0
// ---- helpers
// MFMAs as asm statements, for the hand-interleaved body below: builtin MFMAs are pure values to the optimiser and leave the place
// they were written in (the interleave dissolves before the machine scheduler sees it); a volatile statement stays.  The price: the
// compiler no longer knows these are matrix instructions and inserts none of their wait states --
//   * a VALU instruction may read an MFMA result only ob_fl_settle() later (placed after every phase whose results are read next)
//   * an MFMA may read a VALU result only a few instructions later: every such operand (P, a rescaled accumulator) is
//     materialised where it is made (an empty asm with it as operand) and used a phase later; tests/test_flash_isa.py compiles
//     this header and fails on any VALU write within 8 instructions of an MFMA that reads it.
__device__ __forceinline__ void ob_fl_mfma0(ob_float4 &c, const ob_half8 &a, const ob_half8 &b)
{
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void ob_fl_mfma(ob_float4 &c, const ob_half8 &a, const ob_half8 &b)
{
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void ob_fl_settle() { asm volatile("s_nop 15\n\ts_nop 7"); }

// ---- body
    // Blocks below every diagonal.  Two waves of a SIMD do NOT overlap one's MFMAs with the other's VALU work (measured with an
    // 8-wave ping-pong arrangement, ob_flash_pp.h: MFMA-only time + VALU-only time = total time), only a wave's OWN VALU
    // instructions run in the shadow of its MFMAs.  So the two query tiles of the wave are taken one after the other and the
    // exp2 / row-sum / convert stream of one tile is interleaved with the other tile's MFMAs by sched_group_barrier:
    //   A  scores(q0)                    B  scores(q1) + exponentials(q0)
    //   C  output(q0) + exponentials(q1) D  output(q1)
    // (the maxima and the rescale decision of a tile sit between the phases: they are what the next phase's arithmetic waits
    // for).  K and V fragments are read once per tile, i.e. twice per block -- LDS was 25 % busy.  Same arithmetic in the same
    // order as the plain body, which the diagonal blocks keep.
    auto block_main = [&](const int kb) {
        const int buf = kb & 1;
        const _Float16 *Kb = &Ks[buf][0][0], *Vb = &Vs[buf][0][0];
        store_block(buf ^ 1);
        auto frag_k = [&](int i) {                      // i = 4 (d step) + key tile: 4 independent accumulators between dependent MFMAs
            return *reinterpret_cast<const ob_half8 *>(Kb + 16 * (i % 4) * D + koff[i / 4]);
        };
        auto frag_v = [&](int j) {                      // j = DT (key step) + d tile
            const ob_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ob_lds_v4s *)(Vb + 32 * (j / DT) * D + voff[j % DT]));
            const ob_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ob_lds_v4s *)(Vb + (32 * (j / DT) + 16) * D + voff[j % DT]));
            const ob_v8s a8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(ob_half8, a8);
        };
        constexpr int NF = 4 * DK, NV = 2 * DT, PRE = 4;
        const ob_float4 zero4 = {0.f, 0.f, 0.f, 0.f};
        ob_float4 sc[2][4];
        ob_half8 pb[2][2];
        float mx[2], nm[2];
        float ls[2];
        // one step of a tile's exponentials: value i of its 16 (key tile i / 4, element i % 4); the fp16 P operand of a key step
        // is complete after 8 values
        auto exp_step = [&](int qt, int i) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[qt][i / 4][i % 4], A.scale_log2e, nm[qt]));
            sc[qt][i / 4][i % 4] = p;
            ls[qt] = i == 0 ? p : ls[qt] + p;
            if (i % 8 == 7) {
                const int ks = i / 8;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pb[qt][ks][e] = (_Float16)sc[qt][2 * ks][e];
                    pb[qt][ks][4 + e] = (_Float16)sc[qt][2 * ks + 1][e];
                }
            }
            if (i == 15) {
                l_run[qt] += ls[qt];
                // materialised HERE: left alone, the stream is sunk to its first use, behind the next tile's maxima
                asm volatile("" : "+v"(pb[qt][0]), "+v"(pb[qt][1]), "+v"(l_run[qt]));
            }
        };
        auto maximum = [&](int qt) {                    // row maxima of the tile, and the (rare) move of the running maximum
            float m = sc[qt][0][0];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) m = fmaxf(m, sc[qt][kt][e]);
            m = ob_fl_col_max(m);
            if (__builtin_amdgcn_ballot_w64((m - m_run[qt]) * A.scale_log2e > OB_FL_DEFER_THR) != 0) {
                const float m_new = fmaxf(m_run[qt], m);
                const float alpha = m_new == -INFINITY ? 1.f : __builtin_amdgcn_exp2f((m_run[qt] - m_new) * A.scale_log2e);
                l_run[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    acc_o[dt][qt] *= alpha;
                    asm volatile("" : "+v"(acc_o[dt][qt]));
                }
                m_run[qt] = m_new;
            }
            nm[qt] = m_run[qt] == -INFINITY ? 0.f : -m_run[qt] * A.scale_log2e;
        };
        // a phase = 16 steps of {MFMA, fragment request PRE steps ahead, VALU of the other tile}; nothing crosses a step boundary
        ob_half8 kf[NF], vf[NV];
        auto scores = [&](int qt, int other) {          // other < 0: no softmax work to interleave
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                if (i / 4 == 0) ob_fl_mfma0(sc[qt][i % 4], kf[i], qf[qt][i / 4]);
                else ob_fl_mfma(sc[qt][i % 4], kf[i], qf[qt][i / 4]);
                if (i + PRE < NF) kf[i + PRE] = frag_k(i + PRE);
                if (other >= 0) exp_step(other, i);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto output = [&](int qt, int other) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                ob_fl_mfma(acc_o[j % DT][qt], vf[j], pb[qt][j / DT]);
                if (j + PRE < NV) vf[j + PRE] = frag_v(j + PRE);
                if (other >= 0) exp_step(other, j);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        (void)mx;
        // ---- A
#pragma unroll
        for (int i = 0; i < PRE; ++i) kf[i] = frag_k(i);
        __builtin_amdgcn_sched_barrier(0);
        scores(0, -1);
        ob_fl_settle();
        load_block(kb + 2);
#pragma unroll
        for (int i = 0; i < PRE; ++i) kf[i] = frag_k(i);         // second pass over the K tile: on its way during the maxima
        __builtin_amdgcn_sched_barrier(0);
        maximum(0);
        // ---- B
        scores(1, 0);
        ob_fl_settle();
#pragma unroll
        for (int j = 0; j < PRE; ++j) vf[j] = frag_v(j);
        __builtin_amdgcn_sched_barrier(0);
        maximum(1);
        // ---- C
        output(0, 1);
#pragma unroll
        for (int j = 0; j < PRE; ++j) vf[j] = frag_v(j);
        __builtin_amdgcn_sched_barrier(0);
        // ---- D
        output(1, -1);
        ob_fl_settle();                                 // (the next block, or the epilogue, may be compiler-scheduled code)
        __syncthreads();
    };
#endif
