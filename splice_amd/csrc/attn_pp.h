// Attention forward, "ping-pong" form (round 5; included by attention.hip, which supplies the tile toolkit).
//
// One workgroup = EIGHT waves = two groups of four that own the SAME 64 * QB queries and walk the two halves of the key range
// (as the two-group form before it: same key split, same merge, so a query's arithmetic does not depend on the launch).
// A workgroup's waves w and w + 4 share a SIMD; the two groups run ONE PHASE APART:
//
//      group 0 :  M(0) | V(0) | M(1) | V(1) | ...            M(i) = matrix phase:  P V of tile i-1  +  K Q^T of tile i   (32 MFMAs at
//      group 1 :       | M(0) | V(0) | M(1) | V(1) | ...            QB = 2, nothing else but the LDS-DMA issue of tile i+2)
//                 one s_barrier between phases                V(i) = vector phase:  softmax of tile i (exp2, row sums, bf16 packing)
//                                                                    + the LDS fragment reads of K(i+1) and V(i) into registers
//
// so that on every SIMD one wave feeds the matrix pipe while its partner issues the softmax VALU work and the LDS reads --
// the two pipes overlap by construction instead of by the chance interleaving of four lock-stepped waves
// (/opt/skills/guides/MI355X_MICROARCH.md, "Two waves per SIMD"; the old loop ran wait -> K reads -> MFMA -> softmax -> V reads
// -> MFMA serially in every wave, all eight waves in the same stage: profiles/r04_pmc_attn_selfsim_p1.txt, 46 % of the wave
// cycles parked).  Every MFMA operand is in registers when its phase starts: nothing in a matrix phase waits on LDS.
//
// K / V tiles: [64 keys][64 d] token tiles in a 3-slot LDS ring per group (LDS-DMA issued from inline asm TWO tiles ahead,
// counted vmcnt: a tile has three phases to land), same fragment layouts and swizzle as the other forms (conflict-free).
//
// FOLD: q arrives pre-multiplied by scale * log2(e) (the ViT engine packs the q rows of the QKV projection that way, see
// vit_engine.hip pack_qkv) and the score MFMAs start from the accumulator -m instead of 0: the score IS the exponent, the
// 16 scale-and-shift FMAs per 16 scores are gone.  Without FOLD (the stand-alone C entry point on plain q) p = exp2(fma(s, c, -m)).
#pragma once

constexpr int PP_SLOT_ELEMS = 8192;   // bf16 elements of a ring slot: K tile | V tile
constexpr int PP_SLOTS = 3;
constexpr int PP_LDS_BYTES = 2 * PP_SLOTS * PP_SLOT_ELEMS * 2;   // two groups: 96 KB, one workgroup per CU

__device__ __forceinline__ void pp_dma16(uint32_t lds_byte, uint32_t voff, const void* sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte), "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void pp_wait_dma() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

#ifndef PP_ABL   // timing-only ablations (tools/attn_ablate.sh ... PP_ABL): 1 no softmax, 2 no LDS fragment reads, 4 no DMA in the loop, 8 no MFMAs, 16 no exact-step path
#define PP_ABL 0
#endif
template <class V> __device__ __forceinline__ void pp_opaque(V& v) { asm volatile("" : "+v"(v)); }

// softmax of one 64-key tile for QB blocks of 16 queries: s -> packed bf16 probabilities pb, row sums into l.  Deferred maximum
// as in attn_fwd_tile: probabilities against the reference point of the last exact step, the exact step (per query) when a
// lane's partial row sum leaves [0, 2^30) and on the first tile of a walk.
template <int QB, bool FOLD, bool MASK>
__device__ __forceinline__ void pp_softmax(f32x4 (&s)[QB][4], float (&m)[QB], float (&l)[QB], f32x4 (&negm)[QB], f32x4 (&o)[QB][4],
                                           u32x4 (&pb)[QB][2], float c2, int kt, int T, int g, bool first) {
    if (MASK) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = kt + (nb >> 1) * 32 + g * 8 + (nb & 1) * 4 + r < T;
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) s[qb][nb][r] = valid ? s[qb][nb][r] : NEG_BIG;
            }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {   // each block of 16 queries on its own: 16 probabilities live at a time
        f32x4 p[4];
        float part[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                p[nb][r] = FOLD ? __builtin_amdgcn_exp2f(s[qb][nb][r]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][nb][r], c2, -m[qb]));
            part[nb] = (p[nb][0] + p[nb][1]) + (p[nb][2] + p[nb][3]);
        }
        float ps = (part[0] + part[1]) + (part[2] + part[3]);
        const bool over = first || !(ps < ATTN_RESCALE_LIMIT);
        if (!(PP_ABL & 16) && __any(over)) {
            // exact online-softmax step (rare), taking effect PER QUERY (a query's bits must not depend on its wave mates): only a
            // query one of whose four lanes ran over moves its reference maximum; for the others mn = m, alpha = 1 and the same
            // p and row sum come out again (same summation order as above)
            float mx = NEG_BIG;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qb][nb][r]);
            mx = FOLD ? group4_max(mx) + m[qb] : group4_max(mx) * c2;   // FOLD: scores are held as s - m
            const bool mine = group4_max(over ? 1.0f : 0.0f) > 0.f;
            const float mold = m[qb];
            const float mn = first ? mx : (mine ? fmaxf(mold, mx) : mold);
            const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(mold - mn);   // first tile: o = l = 0
            m[qb] = mn;
            negm[qb] = f32x4{-mn, -mn, -mn, -mn};
            l[qb] *= alpha;
#pragma unroll
            for (int nd = 0; nd < 4; ++nd)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[qb][nd][r] *= alpha;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    p[nb][r] = FOLD ? __builtin_amdgcn_exp2f(s[qb][nb][r] + (mold - mn)) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][nb][r], c2, -mn));
                part[nb] = (p[nb][0] + p[nb][1]) + (p[nb][2] + p[nb][3]);
            }
            ps = (part[0] + part[1]) + (part[2] + part[3]);
        }
        l[qb] += ps;
        pb[qb][0] = pack8v(p[0], p[1]);
        pb[qb][1] = pack8v(p[2], p[3]);
    }
}

#ifndef PP_PRIO
#define PP_PRIO 0
#endif

template <int QB, bool FOLD>
__global__ __launch_bounds__(512) void attn_fwd_pp_kernel(AttnArgs a, int nx) {
    extern __shared__ __attribute__((aligned(16))) bf16_t pp_smem[];
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, c = lane & 15;
#ifndef PP_GRP_MODE
#define PP_GRP_MODE 0
#endif
    const int w8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = PP_GRP_MODE == 0 ? w8 >> 2 : PP_GRP_MODE == 1 ? w8 & 1 : (w8 >> 1) & 1;
    int xb, h, b;
    attn_block_coords(nx, a.H, a.B, xb, h, b);
    const int ld = 3 * a.D;
    TileDmaT<4> dma;
    dma.wave = PP_GRP_MODE == 0 ? w8 & 3 : PP_GRP_MODE == 1 ? w8 >> 1 : (w8 & 1) | ((w8 >> 2) << 1);
    dma.tchunk = (lane & 7) ^ ((((dma.lrow >> 1) & 1) << 1) | ((dma.wave & 1) << 2));
    const FragAddr fa(g, c);
    const int qbase = xb * (64 * QB) + dma.wave * (16 * QB);
    const bool active = qbase < a.Tld;   // a wave without queries still moves its share of every tile
    const bf16_t* qkv_b = a.qkv + (size_t)b * a.Tld * ld;
    const bf16_t* kbase = qkv_b + a.D + h * 64;
    const bf16_t* vbase = qkv_b + 2 * a.D + h * 64;

    // key tiles with at least one valid key: [0, nt); the range is ALWAYS split in two (same bits for every launch form)
    const int nt = (a.T + 63) / 64, per = (nt + 1) / 2;
    const int t0 = grp * per;
    const int ng = max(min(nt, t0 + per) - t0, 0);
    bf16_t* ring = pp_smem + grp * (PP_SLOTS * PP_SLOT_ELEMS);
    const uint32_t ring_b = (uint32_t)(size_t)(__attribute__((address_space(3))) bf16_t*)ring + (uint32_t)dma.wave * 1024u;
    const uint32_t koff = dma.token_off(ld);
    auto issue = [&](int tile, int slot) {   // 4 LDS-DMA instructions per wave: K pieces w, w + 4, V pieces w, w + 4
        const int kt = tile * 64;
        const uint32_t dst = ring_b + (uint32_t)slot * (PP_SLOT_ELEMS * 2);
        if (kt + 64 <= a.Tld) {
            const bf16_t* kb = kbase + (size_t)kt * ld;
            const bf16_t* vb = vbase + (size_t)kt * ld;
            pp_dma16(dst, koff, kb);
            pp_dma16(dst + 4096, koff, kb + (size_t)32 * ld);
            pp_dma16(dst + 8192, koff, vb);
            pp_dma16(dst + 8192 + 4096, koff, vb + (size_t)32 * ld);
        } else {   // edge tile: rows past the pass are clamped duplicates (masked keys)
            uint32_t off[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int t = kt + (dma.wave + 4 * i) * 8 + dma.lrow;
                t = t < a.Tld ? t : a.Tld - 1;
                off[i] = (uint32_t)((t * ld + dma.tchunk * 8) * 2);
            }
            pp_dma16(dst, off[0], kbase);
            pp_dma16(dst + 4096, off[1], kbase);
            pp_dma16(dst + 8192, off[0], vbase);
            pp_dma16(dst + 8192 + 4096, off[1], vbase);
        }
    };
    if (ng > 0) issue(t0, 0);
    if (ng > 1) issue(t0 + 1, 1);

    u32x4 qf[QB][2];
    int qidx[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int q = qbase + qb * 16 + c;
        qidx[qb] = q;
        q = q < a.Tld ? q : a.Tld - 1;
        const bf16_t* p = qkv_b + (size_t)q * ld + h * 64 + g * 8;
        qf[qb][0] = ld16v(p);
        qf[qb][1] = ld16v(p + 32);
    }
    float m[QB], l[QB];
    f32x4 o[QB][4], negm[QB], s[QB][4];
    u32x4 pb[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m[qb] = FOLD ? 0.f : NEG_BIG;
        l[qb] = 0.f;
        negm[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
        pb[qb][0] = pb[qb][1] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) o[qb][nd] = s[qb][nd] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float c2 = a.scale * LOG2E;
    u32x4 kf[4][2], vf[2][4];
    auto read_k = [&](const bf16_t* Ks) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            kf[nb][0] = lds16(Ks + fa.tok[0] + (nb >> 1) * 2048 + (nb & 1) * 256);
            kf[nb][1] = lds16(Ks + fa.tok[1] + (nb >> 1) * 2048 + (nb & 1) * 256);
        }
    };
    auto read_v = [&](const bf16_t* Vs) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int nd = 0; nd < 4; ++nd) vf[sub][nd] = lds_tr16(Vs, fa.tr[nd] + sub * 2048);
    };
    // tiles 0 and 1 (and the queries) have landed for every wave
    pp_wait_dma<0>();
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) asm volatile("" : "+v"(qf[qb][0]), "+v"(qf[qb][1]));   // the compiler's own wait for the q loads sits HERE, not inside the first matrix phase
    pp_barrier();
    if (active && ng > 0) read_k(ring);
    if (grp == 1) pp_barrier();   // group 1 runs one phase behind

    int slot_i = 0, slot_n = 2;   // slot of tile i, slot of tile i + 2
    for (int i = 0; i <= per; ++i) {
        // ---- M(i): P V of tile i-1, K Q^T of tile i
        if (!(PP_ABL & 4) && i + 2 < ng) issue(t0 + i + 2, slot_n);
#if PP_PRIO
        __builtin_amdgcn_s_setprio(PP_PRIO);
#endif
        if ((PP_ABL & 8) && active) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int nd = 0; nd < 4; ++nd) { pp_opaque(o[qb][nd]); pp_opaque(s[qb][nd]); }
        }
        if (!(PP_ABL & 8) && active) {
            if (i >= 1 && i <= ng) {
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int nd = 0; nd < 4; ++nd)
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) o[qb][nd] = mfma16(vf[sub][nd], pb[qb][sub], o[qb][nd]);
            }
            if (i < ng) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) s[qb][nb] = mfma16(kf[nb][0], qf[qb][0], FOLD ? negm[qb] : f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) s[qb][nb] = mfma16(kf[nb][1], qf[qb][1], s[qb][nb]);
            }
        }
#if PP_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if (i + 2 < ng) pp_wait_dma<4>(); else pp_wait_dma<0>();   // tile i + 1 has landed (only tile i + 2 may still fly)
        pp_barrier();
        // ---- V(i): fragment reads for the next matrix phase, softmax of tile i
        if (active && i < ng) {
            const int sn = slot_i + 1 == PP_SLOTS ? 0 : slot_i + 1;
            if (!(PP_ABL & 2)) {
                if (i + 1 < ng) read_k(ring + sn * PP_SLOT_ELEMS);
                read_v(ring + slot_i * PP_SLOT_ELEMS + 4096);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { pp_opaque(kf[j][0]); pp_opaque(kf[j][1]); pp_opaque(vf[0][j]); pp_opaque(vf[1][j]); }
            }
            const int kt = (t0 + i) * 64;
            if (PP_ABL & 1) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) { pp_opaque(pb[qb][0]); pp_opaque(pb[qb][1]); }
            } else if (kt + 64 <= a.T) pp_softmax<QB, FOLD, false>(s, m, l, negm, o, pb, c2, kt, a.T, g, i == 0);
            else pp_softmax<QB, FOLD, true>(s, m, l, negm, o, pb, c2, kt, a.T, g, i == 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads are out of the slots the next phase's DMA refills
        pp_barrier();
        slot_i = slot_i + 1 == PP_SLOTS ? 0 : slot_i + 1;
        slot_n = slot_n + 1 == PP_SLOTS ? 0 : slot_n + 1;
    }
    if (grp == 0) pp_barrier();
    if (ng == 0) {   // (a one-tile pass: group 1 holds the empty state)
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) m[qb] = NEG_BIG;
    }

    // merge the groups' partial states: lane for lane (same query, same output columns in both groups)
    {
        float* ex = reinterpret_cast<float*>(pp_smem);   // [wave4][QB][18][64]
        if (grp == 1) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                float* e = ex + ((dma.wave * QB + qb) * 18) * 64 + lane;
                e[0] = m[qb];
                e[64] = l[qb];
#pragma unroll
                for (int nd = 0; nd < 4; ++nd)
#pragma unroll
                    for (int r = 0; r < 4; ++r) e[(2 + nd * 4 + r) * 64] = o[qb][nd][r];
            }
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float* e = ex + ((dma.wave * QB + qb) * 18) * 64 + lane;
            const float m1 = e[0], l1 = e[64];
            const float mn = fmaxf(m[qb], m1);
            const float a0 = __builtin_amdgcn_exp2f(m[qb] - mn), a1 = __builtin_amdgcn_exp2f(m1 - mn);
            m[qb] = mn;
            l[qb] = merge2(l[qb], a0, l1, a1);
#pragma unroll
            for (int nd = 0; nd < 4; ++nd)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[qb][nd][r] = merge2(o[qb][nd][r], a0, e[(2 + nd * 4 + r) * 64], a1);
        }
    }
    if (!active) return;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float lt = group4_sum(l[qb]);
        const int q = qidx[qb];
        if (q < a.Tld) {
            const float inv = 1.0f / lt;
            bf16_t* op = a.out + ((size_t)b * a.Tld + q) * a.D + h * 64 + g * 4;
#pragma unroll
            for (int nd = 0; nd < 4; ++nd) st4bf(op + nd * 16, o[qb][nd], inv);
            if (g == 0) a.lse[((size_t)b * a.H + h) * a.Tld + q] = m[qb] + __builtin_amdgcn_logf(lt);   // log2 units
        }
    }
}
