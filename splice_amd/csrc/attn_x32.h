// Attention forward on the 32x32x16 MFMA (round 5; included by attention.hip).
//
// Why this shape.  tools/micro/issue_model.hip (profiles/r05_issue_model.txt) measured what bounds a softmax-heavy MFMA loop on
// gfx950: a SIMD has ONE issue port for MFMA and VALU instructions of all its waves.  An MFMA of either shape holds it ~9 cycles,
// a plain VALU 4, v_exp_f32 8, v_cvt_pk_bf16_f32 4.25, v_add_f32 2.3; the matrix pipe runs 16.5 cycles per 16x16x32 and 33 per
// 32x32x16.  Per 32 queries x 64 keys of head dimension 64 that is 525 matrix cycles against a port load of
//      16x16x32, scale-and-shift FMA per score (the round-2..4 kernels):  32*9 + 32*8 + 32*4 + 16*4.25 + sums/moves ~ 190  = 930   -> <= 56 %
//      32x32x16, q pre-scaled (FOLD), row sums as v_add:                  16*9 + 32*8 +        16*4.25 + 32*2.3 + ~40     = 580   -> <= 90 %
// (the measured "MFMA + VALU only" skeleton of the old kernel was 54 %, profiles/r03_attn_ablation.txt -- the same number).  So:
// half the MFMA instructions for the same FLOPs, no FMA per score, no register-pair moves, and waves that run FREELY (one
// s_barrier per key tile for the LDS ring, nothing else in lock step) so that one wave's softmax fills the port while another's
// MFMAs run; several small workgroups per CU instead of one big one (they de-phase on their own).  A lock-stepped two-group
// "ping-pong" (attn_pp.h) was built first and measured: phases of the two groups ADD, they do not overlap
// (profiles/r05_pp_ablate1.txt, tools/micro/pp_overlap.hip) -- kept in the tree as the record of that experiment.
//
// Wave = 32 queries (one per lane & 31; the lane halves hi = lane >> 5 split the keys of a block), tile = 64 keys:
//   S^T[64 keys][32 q]  = 2 blocks x 4 MFMAs   A = K rows (one ds_read_b128 each), B = q (registers), C = -m (FOLD) or 0
//   O^T[64 d][32 q]    += 2 blocks x 4 MFMAs   A = V^T (two ds_read_b64_tr_b16 out of the V token tile), B = packed P
// C layout of the 32x32 MFMA: lane (q, hi), register r holds row 8 (r >> 2) + 4 hi + (r & 3): registers 8k .. 8k+7 of a score
// block are exactly the eight keys {16k + 4hi + 0..3, 16k + 8 + 4hi + 0..3} -- the B operand of the P V MFMA of key step k once
// packed to bf16 (no cross-lane traffic), and V^T is fetched in the SAME key order (two transposing reads 8 rows apart).
//
// LDS: K ring 3 x 8 KB + V ring 3 x 8 KB per workgroup (48 KB: three workgroups per CU).  K(t+3) and V(t+2) are issued at the
// top of tile t (LDS-DMA from inline asm, counted vmcnt): two tiles of flight time.  Source-side swizzles: K chunk c of row r
// at c ^ ((r >> 1) & 7) (conflict-free for the b128 lane groups reading 32 consecutive rows at one chunk), V chunk c at
// c ^ (((r >> 1) & 1) << 2) (conflict-free for the transposing reads: 4 rows x 64 B per half wave).
#pragma once
#include <type_traits>

template <int NW>
struct X32 {
    static constexpr int PIECES = 8 / NW;          // 1-KiB pieces of a 64-row tile per wave
    static constexpr int NL = 2 * PIECES;          // LDS-DMA instructions per wave per issue group (K pieces + V pieces)
    static constexpr int LDS_BYTES = 6 * 8192;
};

// LDS-DMA from inline asm (the compiler may not reorder it against the counted waits), counted vmcnt wait, raw barrier fenced for the scheduler
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
template <class V> __device__ __forceinline__ void pp_opaque(V& v) { asm volatile("" : "+v"(v)); }   // pins a value HERE (IR-level sinking ignores sched_barrier)

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__device__ __forceinline__ f32x16_t mfma32(const u32x4& a, const u32x4& b, const f32x16_t& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float half2_max(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }

// The exact walk behind the deferred maximum (never taken on sane data; tests/test_ops_gpu.py forces it): every lane pair (q, hi = 0 / 1)
// runs a scalar online softmax over the even / odd keys of its query straight from global memory, the halves are merged across the pair.
__device__ __forceinline__ void x32_exact_rows(const AttnArgs& a, const bf16_t* pbase, int b, int h, int qidx, int hi, float c2) {
    const int ld = 3 * a.D;
    const int qc = qidx < a.Tld ? qidx : a.Tld - 1;
    float q[64], o[64];
    {
        const bf16_t* p = pbase + (size_t)qc * ld;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u32x4 v = ld16v(p + i * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) { q[i * 8 + 2 * j] = __uint_as_float(v[j] << 16) * c2; q[i * 8 + 2 * j + 1] = __uint_as_float(v[j] & 0xFFFF0000u) * c2; }
        }
    }
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = 0.f;
    float m = NEG_BIG, l = 0.f;
    for (int key = hi; key < a.T; key += 2) {
        const bf16_t* kp = pbase + (size_t)key * ld + a.D;
        float sc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u32x4 v = ld16v(kp + i * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sc = __builtin_fmaf(q[i * 8 + 2 * j], __uint_as_float(v[j] << 16), sc);
                sc = __builtin_fmaf(q[i * 8 + 2 * j + 1], __uint_as_float(v[j] & 0xFFFF0000u), sc);
            }
        }
        const float mn = fmaxf(m, sc);
        const float alpha = __builtin_amdgcn_exp2f(m - mn), pk = __builtin_amdgcn_exp2f(sc - mn);
        m = mn;
        l = l * alpha + pk;
        const bf16_t* vp = pbase + (size_t)key * ld + 2 * a.D;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u32x4 v = ld16v(vp + i * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[i * 8 + 2 * j] = __builtin_fmaf(pk, __uint_as_float(v[j] << 16), o[i * 8 + 2 * j] * alpha);
                o[i * 8 + 2 * j + 1] = __builtin_fmaf(pk, __uint_as_float(v[j] & 0xFFFF0000u), o[i * 8 + 2 * j + 1] * alpha);
            }
        }
    }
    const float m1 = __shfl_xor(m, 32, 64), l1 = __shfl_xor(l, 32, 64);
    const float mn = fmaxf(m, m1);
    const float a0 = __builtin_amdgcn_exp2f(m - mn), a1 = __builtin_amdgcn_exp2f(m1 - mn);
    const float lt = l * a0 + l1 * a1, inv = 1.0f / lt;
    bf16_t* op = a.out + ((size_t)b * a.Tld + qc) * a.D + h * 64;
#pragma unroll
    for (int d = 0; d < 64; ++d) {
        const float od = (o[d] * a0 + __shfl_xor(o[d], 32, 64) * a1) * inv;
        if (hi == (d >> 5) && qidx < a.Tld) op[d] = f2bf(od);   // each lane of the pair stores one half of the row
    }
    if (hi == 0 && qidx < a.Tld) a.lse[((size_t)b * a.H + h) * a.Tld + qidx] = mn + __builtin_amdgcn_logf(lt);
}

template <int NW, bool FOLD>
__global__ __launch_bounds__(64 * NW) void attn_fwd_x32_kernel(AttnArgs a, int nx) {
    extern __shared__ __attribute__((aligned(16))) bf16_t x32_smem[];   // K slots 0..2 | V slots 0..2, 4096 elements each
    constexpr int PIECES = X32<NW>::PIECES, NL = X32<NW>::NL;
    const int lane = threadIdx.x & 63;
    const int q32 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int xb, h, b;
    attn_block_coords(nx, a.H, a.B, xb, h, b);
    const int ld = 3 * a.D;
    const int qbase = xb * (32 * NW) + wave * 32;
    const bool active = qbase < a.Tld;   // a wave without queries still moves its share of every tile
    const bf16_t* pbase = a.qkv + (size_t)b * a.Tld * ld + h * 64;   // this (pass, head): q columns; k at + D, v at + 2 D
    const int nt = (a.T + 63) / 64;

    // ---- LDS-DMA: piece p = wave + NW * i covers tile rows 8p .. 8p+7; lane -> (row 8p + (lane >> 3), position lane & 7).
    // ONE running scalar pointer (the q column of the first row of tile t + 2) serves both streams: the column deltas and the
    // two-tile lead of the K stream sit in the per-lane byte offsets.
    const int lr = lane >> 3, pos = lane & 7;
    const int kchunk = pos ^ (((wave & 1) << 2) | (lr >> 1));          // K: position holds source chunk pos ^ ((row >> 1) & 7); piece & 1 == wave & 1
    const int vchunk = pos ^ (((lr >> 1) & 1) << 2);                   // V: pos ^ (((row >> 1) & 1) << 2)
    uint32_t koffs[PIECES], voffs[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int row = (wave + NW * i) * 8 + lr;
        koffs[i] = (uint32_t)((row * ld + kchunk * 8 + a.D) * 2);
        voffs[i] = (uint32_t)((row * ld + vchunk * 8 + 2 * a.D) * 2);
    }
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) bf16_t*)x32_smem + (uint32_t)wave * 1024u;
    auto issue_tile = [&](const uint32_t (&offs)[PIECES], int chunk, int coldelta, int tile, uint32_t dst) __attribute__((always_inline)) {   // PIECES instructions
        const int kt = tile * 64;
        if (kt + 64 <= a.Tld) {
            const bf16_t* tb = pbase + (size_t)kt * ld;
#pragma unroll
            for (int i = 0; i < PIECES; ++i) pp_dma16(dst + i * (NW * 1024), offs[i], tb);
        } else {   // edge tile: rows past the pass are clamped duplicates (masked keys)
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                int t = kt + (wave + NW * i) * 8 + lr;
                t = t < a.Tld ? t : a.Tld - 1;
                pp_dma16(dst + i * (NW * 1024), (uint32_t)((t * ld + chunk * 8 + coldelta) * 2), pbase);
            }
        }
    };
    auto issue_k = [&](int tile, int slot) __attribute__((always_inline)) { issue_tile(koffs, kchunk, a.D, tile, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)slot * 8192u)); };
    auto issue_v = [&](int tile, int slot) __attribute__((always_inline)) { issue_tile(voffs, vchunk, 2 * a.D, tile, __builtin_amdgcn_readfirstlane(lds0 + 3 * 8192u + (uint32_t)slot * 8192u)); };
    // main-loop form: no conditions, a running scalar pointer per stream (first row, q column, of the tile to fetch)
    const char* kptr = reinterpret_cast<const char*>(pbase) + (size_t)4 * 64 * ld * 2;   // tile t + 4 at the top of tile t
    const char* vptr = reinterpret_cast<const char*>(pbase) + (size_t)2 * 64 * ld * 2;   // tile t + 2
    const size_t tile_bytes = (size_t)64 * ld * 2;
    auto issue_fast = [&](int s1, int s2) __attribute__((always_inline)) {
        const uint32_t kd = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)s1 * 8192u), vd = __builtin_amdgcn_readfirstlane(lds0 + 3 * 8192u + (uint32_t)s2 * 8192u);
#pragma unroll
        for (int i = 0; i < PIECES; ++i) pp_dma16(kd + i * (NW * 1024), koffs[i], kptr);
#pragma unroll
        for (int i = 0; i < PIECES; ++i) pp_dma16(vd + i * (NW * 1024), voffs[i], vptr);
    };
    issue_k(0, 0);
    issue_v(0, 0);
    if (nt > 1) { issue_k(1, 1); issue_v(1, 1); }
    if (nt > 2) issue_k(2, 2);

    // ---- q: B operand of the score MFMAs, lane (q, hi) holds d = 16 ks + 8 hi .. + 7 for ks = 0..3
    u32x4 qf[4];
    const int qidx = qbase + q32;
    {
        const int qc = qidx < a.Tld ? qidx : a.Tld - 1;
        const bf16_t* p = pbase + (size_t)qc * ld + hi * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = ld16v(p + ks * 16);
    }
    // ---- fragment addresses (bf16 elements inside a tile; everything else is an immediate)
    int ka[4];   // K tile: row q32 (+ 32 kb), chunk 2 ks + hi
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ka[ks] = q32 * 64 + (((2 * ks + hi) ^ ((q32 >> 1) & 7)) << 3);
    // V tile, transposing read: 16-lane group (lane >> 4) & 1 takes d 16 .. 31 of the block; lane m of a group points at row 4 hi + (m >> 2),
    // d piece m & 3 (4 values): chunk 4 db + 2 ((lane >> 4) & 1) + ((m >> 1) & 1), half m & 1; swizzle bit = row bit 1 = (m >> 3) & 1
    int va[2];
    {
        const int m = lane & 15, dg = (lane >> 4) & 1;
#pragma unroll
        for (int db = 0; db < 2; ++db)
            va[db] = (4 * hi + (m >> 2)) * 64 + (((4 * db + 2 * dg + ((m >> 1) & 1)) ^ (((m >> 3) & 1) << 2)) << 3) + (m & 1) * 4;
    }
    float m_ref = 0.f, l = 0.f;   // reference point of the probabilities (log2 units), this lane's partial row sum
    f32x16_t o[2], sb[2][2], negm;   // sb[parity of the tile][key block]
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; sb[0][0][r] = sb[0][1][r] = sb[1][0][r] = sb[1][1][r] = 0.f; negm[r] = 0.f; }
    const float c2 = a.scale * LOG2E;
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    u32x4 kf[2][4];
    auto read_k = [&](const bf16_t* Ks) __attribute__((always_inline)) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[kb][ks] = lds16(Ks + ka[ks] + kb * 2048);
    };
    auto scores = [&](f32x16_t (&s)[2], const f32x16_t& seed) __attribute__((always_inline)) {   // S^T = K Q^T + seed from the K fragments in registers
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) s[kb] = mfma32(kf[kb][0], qf[0], seed);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) s[kb] = mfma32(kf[kb][ks], qf[ks], s[kb]);
    };
    const bf16_t* Kring = x32_smem;
    const bf16_t* Vring = x32_smem + 3 * 4096;

    // ---- prologue: the reference point = the exact row maximum of tile 0; S(0) from it; K(1) into registers
    pp_wait_dma<0>();
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));   // the compiler's wait for the q loads sits here
    pp_barrier();
    if (active) {
        read_k(Kring);
        scores(sb[0], zero16);
        float mx = NEG_BIG;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool valid = 32 * kb + 8 * (r >> 2) + 4 * hi + (r & 3) < a.T;
                mx = fmaxf(mx, valid ? sb[0][kb][r] : NEG_BIG);
            }
        m_ref = half2_max(mx) * (FOLD ? 1.0f : c2);
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m_ref;
        if (FOLD) scores(sb[0], negm);
        if (nt > 1) read_k(Kring + 4096);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every wave holds K(0), K(1) in registers before their slots are refilled
    pp_barrier();
    if (nt > 3) issue_k(3, 0);

    // Tile t: [wait: K(t+2), V(t) landed; barrier] -> issue K(t+4), V(t+2) -> SIXTEEN MFMA slots, each MFMA followed by its share of the
    // tile's other work (a 32x32 MFMA runs 33 cycles and holds the issue port 9: ~24 cycles of VALU / LDS issue hide behind it, but
    // only behind an MFMA of the SAME wave, tools/micro/pp_overlap.hip):
    //   slots 1-8   S(t+1) = K(t+1) Q^T (registers)   |  exp2 / row sums / bf16 packing of tile t, chunks 0-2;  V(t)^T fragment reads
    //   slots 9-16  O += V(t)^T P(t)                   |  chunk 3;  K(t+2) fragment reads
    // Deferred maximum, lazily: the probabilities of tile t are formed against the reference point in force; if a lane's row sum of
    // the tile reaches 2^30 the state (O, l, the scores of tile t+1 already under way) is rescaled AFTER the tile (all of it is
    // relative to the same point, so this is exact as long as no single probability overflowed fp32: a score would have to exceed
    // everything seen in earlier tiles by 2^97 inside one tile; such a query ends non-finite and is recomputed by the exact walk
    // at the end of the kernel).
    int slot = 0;   // t % 3
#define X32_SB __builtin_amdgcn_sched_barrier(0)
    auto body = [&](auto PARC, auto TAILC, int t) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PARC)::value;
        constexpr bool TAIL = decltype(TAILC)::value;   // main-loop tiles (t + 5 < nt): every fetch exists and is a full tile, one wait form
        f32x16_t (&s)[2] = sb[PAR];
        f32x16_t (&sn)[2] = sb[PAR ^ 1];
        const int s1 = slot == 2 ? 0 : slot + 1, s2 = slot == 0 ? 2 : slot - 1;   // (t + 1) % 3, (t + 2) % 3
        if (!TAIL || t + 3 < nt) pp_wait_dma<NL>(); else pp_wait_dma<0>();   // only the group issued last tile may still fly
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // this wave's fragment reads are out of the slots refilled below
        pp_barrier();
        if (!TAIL) issue_fast(s1, s2);
        else {
            if (t + 4 < nt) issue_k(t + 4, s1);   // K(t+1) left its slot for the registers during tile t - 1
            if (t + 2 < nt) issue_v(t + 2, s2);   // V(t-1) was consumed by tile t - 1
        }
        kptr += tile_bytes;
        vptr += tile_bytes;
        if (active) {
            const bf16_t* Vs = Vring + slot * 4096;
            const bf16_t* Kn = Kring + s2 * 4096;
            u32x4 vf[4][2], pb[4];   // V^T fragments [key step][d block], packed probabilities [key step]
            float p[2][16];
            float pa[4] = {0.f, 0.f, 0.f, 0.f};   // four independent partial row sums
            const int kt = t * 64;
            if (TAIL && kt + 64 > a.T) {   // last tile: padding keys
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[kb][r] = kt + 32 * kb + 8 * (r >> 2) + 4 * hi + (r & 3) < a.T ? s[kb][r] : NEG_BIG;
            }
            auto ex = [&](int kb, int r0, int n) __attribute__((always_inline)) {
#pragma unroll
                for (int r = r0; r < r0 + n; ++r) {
                    p[kb][r] = FOLD ? __builtin_amdgcn_exp2f(s[kb][r]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], c2, -m_ref));
                    pa[r & 3] += p[kb][r];
                }
                asm volatile("" : "+v"(pa[0]), "+v"(pa[1]), "+v"(pa[2]), "+v"(pa[3]));   // (pins the adds HERE: IR-level sinking ignores sched_barrier)
            };
            auto cv = [&](int kk) __attribute__((always_inline)) {
                const float* pp = &p[kk >> 1][(kk & 1) * 8];
                pb[kk] = u32x4{pack2bf(pp[0], pp[1]), pack2bf(pp[2], pp[3]), pack2bf(pp[4], pp[5]), pack2bf(pp[6], pp[7])};
                pp_opaque(pb[kk]);   // (pins the conversions HERE)
            };
            auto rv = [&](int kk, int db) __attribute__((always_inline)) {
                typedef __attribute__((address_space(3))) tr_v4s lds_v4s;
                const tr_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(Vs + va[db] + kk * 1024));
                const tr_v4s hh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(Vs + va[db] + kk * 1024 + 512));
                const uint2 x = __builtin_bit_cast(uint2, lo), y = __builtin_bit_cast(uint2, hh);
                vf[kk][db] = u32x4{x.x, x.y, y.x, y.y};
            };
            auto rk = [&](int kb, int ks) __attribute__((always_inline)) {
                kf[kb][ks] = lds16(Kn + ka[ks] + kb * 2048);   // (past the last tile: stale bytes, never used)
            };
            auto qk = [&](int kb, int ks) __attribute__((always_inline)) {
                // (past the last tile: a dummy product).  The first MFMA of a chain takes its accumulator from -m and writes a DIFFERENT register block:
                // through the builtin the compiler copies the sixteen registers first (two-address form), 16 v_mov_b64 per tile.
                if (ks == 0 && FOLD) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(sn[kb]) : "v"(kf[kb][0]), "v"(qf[0]), "v"(negm));
                else sn[kb] = mfma32(kf[kb][ks], qf[ks], ks == 0 ? zero16 : sn[kb]);
            };
            auto pv = [&](int kk, int db) __attribute__((always_inline)) {
                o[db] = mfma32(vf[kk][db], pb[kk], o[db]);
            };
            X32_SB;
            qk(0, 0); X32_SB; ex(0, 0, 3); rv(0, 0); X32_SB;
            qk(1, 0); X32_SB; ex(0, 3, 3); rv(0, 1); X32_SB;
            qk(0, 1); X32_SB; ex(0, 6, 2); cv(0); rv(1, 0); X32_SB;
            qk(1, 1); X32_SB; ex(0, 8, 3); rv(1, 1); X32_SB;
            qk(0, 2); X32_SB; ex(0, 11, 3); rv(2, 0); X32_SB;
            qk(1, 2); X32_SB; ex(0, 14, 2); cv(1); rv(2, 1); X32_SB;
            qk(0, 3); X32_SB; ex(1, 0, 3); rv(3, 0); X32_SB;
            qk(1, 3); X32_SB; ex(1, 3, 3); rv(3, 1); X32_SB;
            pv(0, 0); X32_SB; ex(1, 6, 2); cv(2); rk(0, 0); X32_SB;
            pv(0, 1); X32_SB; ex(1, 8, 3); rk(1, 0); X32_SB;
            pv(1, 0); X32_SB; ex(1, 11, 3); rk(0, 1); X32_SB;
            pv(1, 1); X32_SB; ex(1, 14, 2); cv(3); rk(1, 1); X32_SB;
            pv(2, 0); X32_SB; rk(0, 2); X32_SB;
            pv(2, 1); X32_SB; rk(1, 2); X32_SB;
            pv(3, 0); X32_SB; rk(0, 3); X32_SB;
            pv(3, 1); X32_SB; rk(1, 3); X32_SB;
            const float ps = (pa[0] + pa[1]) + (pa[2] + pa[3]);
            l += ps;
            if (__builtin_expect(__any(!(ps < ATTN_RESCALE_LIMIT)), 0)) {
                // lazy exact step, per QUERY: new reference point = old + log2(row sum so far); everything accumulated so far, and the scores of
                // tile t + 1 already under way, move to it
                const bool mine = half2_max(!(ps < ATTN_RESCALE_LIMIT) ? 1.0f : 0.0f) > 0.f;
                const float lq = l + __shfl_xor(l, 32, 64);
                const float shift = mine ? __builtin_amdgcn_logf(lq) : 0.f;   // log2
                const float alpha = __builtin_amdgcn_exp2f(-shift);
                m_ref += shift;
                l *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    negm[r] = -m_ref; o[0][r] *= alpha; o[1][r] *= alpha;
                    if (FOLD) { sn[0][r] -= shift; sn[1][r] -= shift; }
                }
            }
        }
        slot = s1;
    };
    {
        std::integral_constant<int, 0> P0; std::integral_constant<int, 1> P1;
        std::false_type MAIN; std::true_type TAILT;
        int t = 0;
        for (; t + 6 < nt; t += 2) {   // tiles t and t + 1 both satisfy t' + 5 < nt
            body(P0, MAIN, t);
            body(P1, MAIN, t + 1);
        }
        for (; t < nt; t += 2) {
            body(P0, TAILT, t);
            if (t + 1 < nt) body(P1, TAILT, t + 1);
        }
    }
#undef X32_SB
    if (!active) return;
    // ---- normalise and store: lane (q, hi) holds O[q][32 db + 8 j + 4 hi + 0..3], j = r >> 2
    const float lt = l + __shfl_xor(l, 32, 64);
    // A healthy row sum is >= 1 (the largest probability since the last reference move is 1) and far from overflow.  Anything else -- inf / NaN
    // (a probability overflowed between two looks at the row sums) or 0 (a rescale factor below 2^-126 flushed the state) -- takes the exact walk.
    if (__builtin_expect(__any(!(lt > 0.25f && lt < 1.0e37f)), 0)) {
        x32_exact_rows(a, pbase, b, h, qidx, hi, FOLD ? 1.0f : c2);
        return;
    }
    if (qidx < a.Tld) {
        const float inv = 1.0f / lt;
        bf16_t* op = a.out + ((size_t)b * a.Tld + qidx) * a.D + h * 64 + hi * 4;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<uint2*>(op + db * 32 + j * 8) =
                    uint2{pack2bf(o[db][4 * j] * inv, o[db][4 * j + 1] * inv), pack2bf(o[db][4 * j + 2] * inv, o[db][4 * j + 3] * inv)};
        if (hi == 0) a.lse[((size_t)b * a.H + h) * a.Tld + qidx] = m_ref + __builtin_amdgcn_logf(lt);   // log2 units
    }
}
