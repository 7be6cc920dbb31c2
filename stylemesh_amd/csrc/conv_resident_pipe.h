// K3p (round 6): the resident-input kernel of conv_split_kernel.h (RES) as a PERSISTENT, CROSS-TILE PIPELINED kernel,
// for the large launches with 64 output channels (conv1_2 forward / data gradient, conv2_1's data gradient at c3 sizes).
//
// The one-tile-per-block kernel spends 3.3 us of a block's 18 - 26 us life in MFMAs (profiles/r05/resident_kernel.txt):
// kernel arguments -> list entry -> 16 loads per thread -> conversion -> barrier -> 36 stages -> epilogue -> store drain
// -> next block's dispatch is one latency chain per tile, covered only by the two other blocks of the CU. Here a block
// stays resident and walks tiles it claims from a per-XCD counter; its input window is staged in HALF PHASES of 32
// channels (two 16-channel chunks = 18 MFMA stages) through two LDS buffers of 8 x 204 units (2 x 25.5 KB - the LDS of the
// one-tile kernel):
//
//     issue the global loads of unit u + 1 (8 loads per thread)  |  loop over unit u (18 stages)  |  convert + store
//     the registers of unit u + 1 -> the other buffer  |  barrier  |  (last unit of a tile: the epilogue)
//
// so the loads of a unit have a whole unit's loop to arrive, the list entry and the
// problem record of tile t + 1 are fetched while tile t computes, the weight ring never drains (every tile reads the same
// image: the last stages of a tile prefetch the first of the next) and no store drain / dispatch separates two tiles.
// The tile after next is claimed with ONE atomic per tile, issued at the start of an epilogue and consumed a tile later.
// SM_EPI_GRAM (conv1_2's data gradient: the epilogue that is HALF of that launch, profiles/r06/respipe_ablation.txt): the
// Gram operand F of the tile travels through the same pipeline as two more units of 32 channels (16-byte loads, 16 KB of a
// buffer each), the derivative matrices' fragments through the weight ring, so that its loads, too, arrive under a loop.
// Chunk / tap / product order of every accumulator = the resident kernel's = the ring kernel's: the sums have its bits
// (tests/test_resident_gpu.py runs both).
#pragma once
#include "conv_split_kernel.h"

namespace sm {

// a wave-uniform float into an SGPR
__device__ __forceinline__ float uniform_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

constexpr int SM_PIPE_HB = 8 * SM_RES_RP;                      // 16-byte units of one half-phase buffer: [chunk][part][k-group][RP]
constexpr int SM_PIPE_SLOTS_PER_CU = 3;
// + bias + claim slot (+ the gate bits of a tile's 64 x 128 Gram operand)
constexpr size_t conv_respipe_lds_bytes(bool gram) { return (size_t)(2 * SM_PIPE_HB) * 16 + 64 * 4 + 16 + (gram ? 8 * 128 : 0); }
constexpr int SM_PIPE_COUNTER_STRIDE = 16;                     // words between the XCDs' tile counters (64 bytes)
constexpr int SM_PIPE_COUNTER_WORDS = 8 * SM_PIPE_COUNTER_STRIDE;

template <int FLAGS, bool UNPOOL>
#ifndef SM_PIPE_WAVES
#define SM_PIPE_WAVES 3
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SM_PIPE_WAVES, SM_PIPE_WAVES)))
void conv3x3_respipe_kernel(ConvArgs a) {
    constexpr int BM = 64, BN = 128, WGM = 2, WGN = 2, NP = SM_SPLIT_NP, NJ = 2, KC = 16, SEG = 4, SEGP = 34, AD = 3;
    constexpr int RP = SM_RES_RP, HB = SM_PIPE_HB;
    static_assert(NP == 2 && 9 % AD == 0, "fp16x2; ring slot of a stage is the same in every chunk");
    constexpr bool GRAM = (FLAGS & SM_EPI_GRAM) != 0;
    static_assert((size_t)(2 * 4 * BN) * 16 <= (size_t)HB * 16, "a 32-channel unit of the Gram operand fits a buffer");
    static_assert(!GRAM || (UNPOOL && FLAGS == (SM_EPI_RELU_MASK | SM_EPI_GRAM)), "the Gram term belongs to conv1_2's data gradient");
    extern __shared__ __attribute__((aligned(16))) f32x4 smem4[];
    f32x4* const Rs = smem4;                                                    // [2 buffers][HB]
    float* const bias_s = reinterpret_cast<float*>(smem4 + 2 * HB);             // [BM]
    int* const slot = reinterpret_cast<int*>(bias_s + BM);                      // the claimed tile, thread 0 -> block
    unsigned char* const Gb = reinterpret_cast<unsigned char*>(slot + 4);       // [8 channel groups][BN] gate bits (GRAM)

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int lhi = lane >> 5;
    const int wm = (wave / WGN) * 32;
    const int wn = (wave % WGN) * (32 * NJ);
    const int n_chunks = a.Cin_pad / KC;
    const int nh = a.Cin_pad / 32;                                              // units (half phases) per tile: 2 or 4

    // ---- tiles: XCD x (= block id mod 8: observed, speed only) owns the contiguous range [start_x, start_x + cnt_x) of the
    // list, as xcd_linear deals it to the one-tile kernel's blocks. A block's FIRST tile is static (sequence number
    // block id / 8 - the grid has at most cnt_x blocks per XCD), every later one a claim on the XCD's counter.
    const int xcd = (int)blockIdx.x & 7;
    const int tq = a.n_tiles >> 3, tr = a.n_tiles & 7;
    const int cnt_x = tq + (xcd < tr ? 1 : 0);
    const int start_x = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int g8 = (int)gridDim.x >> 3;
    unsigned* const counter = a.tile_counter + xcd * SM_PIPE_COUNTER_STRIDE;
    unsigned claimed = 0u;                                                      // (thread 0)

    // ---- weights: as the ring kernel - the global stage image is the MFMA A-fragment layout, a register ring AD stages deep
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wt), 0, 9 * n_chunks * 2 * NP * a.Cout * 16, 0x00020000);
    const int a_voff = (lhi * a.Cout + wm + l31) * 16;
    const int a_part = 2 * a.Cout * 16;
    const int a_stage_bytes = 2 * NP * a.Cout * 16;
    f32x4 ra[AD][NP];
#define SM_LOAD_A(tap_, chunk_)                                                                          \
    {                                                                                                    \
        const int so_ = ((tap_) * n_chunks + (chunk_)) * a_stage_bytes;                                  \
        _Pragma("unroll") for (int s = 0; s < NP; ++s)                                                   \
            ra[(tap_) % AD][s] = __builtin_bit_cast(                                                     \
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, a_voff, so_ + s * a_part, 0));      \
    }
    float inv_in;
    const float in_scale = pow2_scale_for(a.amax_in ? uniform_f(amax_read(a.amax_in)) : 1.f, inv_in);
    const float out_scale = inv_in * a.w_scale_inv;
    const float amax_seen = amax_peek(a.amax_out);
    if ((FLAGS & SM_EPI_BIAS_RELU) && tid < BM) bias_s[tid] = a.bias[tid];

    // ---- staging task of this thread: (group of 8 channels, window row r, block of four consecutive positions) of a unit -
    // 4 x 6 x 9 = 216 tasks, one per thread (the other 40 threads repeat a task's loads and store nothing)
    // (derived from the thread id WHERE it is used - once per tile or unit - instead of living in registers through the loops:
    // the asm makes the id opaque, so that the compiler does not hoist the derivation back out)
    constexpr int RCB = 9, RT = 4 * SM_RES_ROWS * RCB;
    struct Task {
        bool on;
        int grp, row, cb, p0, dst;
    };
    auto window_task = [&]() {
        int t = (int)threadIdx.x;
        asm volatile("" : "+v"(t));
        Task k;
        k.on = t < RT;
        const int tt = k.on ? t : t - RT;
        k.grp = tt / (SM_RES_ROWS * RCB);
        const int rem = tt - k.grp * (SM_RES_ROWS * RCB);
        k.row = rem / RCB;
        k.cb = rem - k.row * RCB;
        k.p0 = UNPOOL ? 4 * k.cb - 1 : 4 * k.cb;                                // window position of the task's element 0
        k.dst = ((k.grp >> 1) * 4 + (k.grp & 1)) * RP + k.row * SEGP + k.p0;    // + j + part * 2 * RP (+ buffer * HB)
        return k;
    };

    // what staging needs of a tile (block-uniform but for src / code / ok)
    struct Stage {
        __amdgpu_buffer_rsrc_t rsrc;
        __amdgpu_buffer_rsrc_t code_rsrc[UNPOOL ? 1 : 0];
        int cstride;                 // bytes between the planes of two channels of the staged tensor
        int src, code;               // byte offsets of the task's first element / code word (channel group 0 of unit 0)
        int ypar;
        bool ok0, ok1;
    };
    // the list entry of a tile: four segments of one problem
    struct Entry {
        int e[SEG];
    };
    auto load_entry = [&](int tile) {
        Entry en;
        const int* e = a.tile_list + (size_t)tile * SEG;
#pragma unroll
        for (int i = 0; i < SEG; ++i) en.e[i] = e[i];
        return en;
    };
    auto problem_of = [&](const Entry& en) {
        ConvProblem P = a.p[0];
        const int gsel = en.e[0] >> 24;
#pragma unroll
        for (int g = 1; g < SM_MAX_GROUP; ++g)
            if (g == gsel) P = a.p[g];
        return P;
    };
    auto make_stage = [&](const Entry& en) {
        const ConvProblem P = problem_of(en);
        const Task tk = window_task();
        const int grp = tk.grp, r_row = tk.row, r_cb = tk.cb;
        const int q0 = en.e[0] & 0xFFFFFF;                                      // first position of the quad (never a padding entry)
        Stage st;
        if constexpr (UNPOOL) {
            const int up_Ho = P.H >> 1, up_Wo = P.W >> 1, up_Wp = row_stride(up_Wo), up_plane = plane_size(up_Ho, up_Wo);
            st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in), 0, 0x7ffffff0, 0x00020000);
            st.code_rsrc[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(P.code), 0, 0x7ffffff0, 0x00020000);
            st.cstride = up_plane * 4;
            // (the resident kernel's un-pooling task: the two pooled elements under image columns x .. x + 3 of image row y)
            const int y0 = q0 / P.Wp - 1, x0 = q0 - (y0 + 1) * P.Wp - 1;
            const int y = y0 - 1 + r_row, x = x0 - 2 + 4 * r_cb;
            const bool yok = (unsigned)y < (unsigned)(2 * up_Ho);
            st.ok0 = yok && (unsigned)x < (unsigned)(2 * up_Wo);
            st.ok1 = yok && (unsigned)(x + 2) < (unsigned)(2 * up_Wo);
            const int yp = min(max((y >> 1) + 1, 0), up_Ho + 1), xp = min(max((x >> 1) + 1, 0), up_Wp - 2);
            const int off_ = yp * up_Wp + xp;
            st.src = (off_ + grp * 8 * up_plane) * 4;
            st.code = (off_ + grp * up_plane) * 4;
            st.ypar = (y & 1) << 1;
        } else {
            st.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.in) - P.Wp - 1, 0, 0x7ffffff0, 0x00020000);
            st.cstride = P.plane * 4;
            const int r_max = P.H + 1 - (q0 / P.Wp - 1);
            st.src = (grp * 8 * P.plane + q0 + 4 * r_cb + min(r_row, r_max) * P.Wp) * 4;
            st.code = 0;
            st.ypar = 0;
            st.ok0 = st.ok1 = true;
        }
        return st;
    };

    // ---- the registers of the unit in flight
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    f32x4 rb[UNPOOL ? 1 : 8];
    f32x2_ rb2[UNPOOL ? 8 : 1];
    u32x2_ rc = {0u, 0u};
    auto issue = [&](const Stage& st, int h) {
        const int so_ = h * 32 * st.cstride;
        if constexpr (UNPOOL) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                rb2[c] = __builtin_bit_cast(f32x2_, __builtin_amdgcn_raw_buffer_load_b64(st.rsrc, st.src, so_ + c * st.cstride, 0));
            rc = __builtin_bit_cast(u32x2_, __builtin_amdgcn_raw_buffer_load_b64(st.code_rsrc[0], st.code, h * 4 * st.cstride, 0));
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                rb[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(st.rsrc, st.src, so_ + c * st.cstride, 0));
        }
    };
    auto convert_store = [&](const Stage& st, int buf) {
        const Task tk = window_task();
        const bool r_on = tk.on;
        const int r_p0 = tk.p0;
        f32x4* const d = Rs + buf * HB + tk.dst;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f16x8 vh, vl;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float v;
                if constexpr (UNPOOL) {          // position r_p0 + j: pooled element j >> 1, window parity (y, j & 1)
                    const int e = j >> 1;
                    const unsigned cw = e ? rc[1] : rc[0];
                    v = e ? rb2[c][1] : rb2[c][0];
                    v = ((e ? st.ok1 : st.ok0) && (int)((cw >> (4 * c)) & 15u) == (st.ypar | (j & 1))) ? v : 0.f;
                } else {
                    v = rb[c][j];
                }
                const float xs_ = __builtin_amdgcn_fmed3f(v * in_scale, -SM_F16_CLAMP, SM_F16_CLAMP);
                const _Float16 h_ = (_Float16)xs_;
                vh[c] = h_;
                vl[c] = (_Float16)(xs_ - (float)h_);
            }
            if (r_on && (unsigned)(r_p0 + j) < (unsigned)SEGP) {
                d[j] = __builtin_bit_cast(f32x4, vh);
                d[j + 2 * RP] = __builtin_bit_cast(f32x4, vl);
            }
        }
    };

    // ---- GRAM: the operand F (= the ReLU gate of the output) of a tile as two units of 32 channels. Task of a thread: four
    // channels (half a k-group) x four consecutive positions of one segment - 4 k-groups x 4 rows x 8 column blocks x 2 halves;
    // the two halves of a k-group are neighbouring lanes. What the Gram units need of a tile:
    struct GStage {
        __amdgpu_buffer_rsrc_t f_rsrc, d_rsrc;     // F planes; the operand images of D0 / D1 (gram_d_pack_group_kernel)
        int cstride;                               // bytes between two channel planes of F
        int src;                                   // byte offset of the task's first element (channel group 0 of unit 0)
        float fscale, oscale;                      // operand scale of F; 1 / (scale of F x scale of D)
        float mk[NJ][2];                           // the lane's mask values of its two column tiles
        bool anyk[NJ][2], alive[NJ];
    };
    struct GTask {
        int half, kg, row, cb, pos;
    };
    auto gram_task = [&]() {
        int t = (int)threadIdx.x;
        asm volatile("" : "+v"(t));
        GTask k;
        k.half = t & 1;
        const int task = t >> 1;
        k.kg = task >> 5;
        k.row = (task >> 3) & 3;
        k.cb = task & 7;
        k.pos = k.row * 32 + 4 * k.cb;                                          // position of the block, element 0 of the task
        return k;
    };
    const int d_voff = (lhi * BM + wm + l31) * 16;                              // this lane's unit of a derivative-matrix fragment
    auto make_gstage = [&](const Entry& en) {
        GStage gs;
        if constexpr (GRAM) {
            const ConvProblem P = problem_of(en);
            const GTask gk = gram_task();
            const int g_half = gk.half, g_kg = gk.kg, g_row = gk.row, g_cb = gk.cb;
            const int q_end = (P.H + 1) * P.Wp;
            gs.f_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.gate), 0, 0x7ffffff0, 0x00020000);
            gs.d_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(P.gram_p), 0, 2 * (6 * BM * BM / 16) * 16, 0x00020000);
            gs.cstride = P.plane * 4;
            const int s0 = en.e[0] & 0xFFFFFF;
            int q_seg = s0;
#pragma unroll
            for (int k = 1; k < SEG; ++k)
                if (g_row == k && (en.e[k] & 0xFFFFFF) != 0xFFFFFF) q_seg = en.e[k] & 0xFFFFFF;
            // (a run that passes the end of its row or plane reads on - into the next rows, the next channel's plane or the
            // buffer's guard floats: those positions carry no mask value and are not stored)
            const int q = q_seg + 4 * g_cb;
            gs.src = ((g_kg * 8 + g_half * 4) * P.plane + q) * 4;
            float inv_f, inv_d;
            gs.fscale = conv_gram_pow2_scale(uniform_f(amax_read(P.gram_amax_feat)), inv_f);
            conv_gram_pow2_scale(uniform_f(amax_read(P.gram_amax_d)), inv_d);
            gs.oscale = inv_f * inv_d;
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj) {
                int qn = s0;
                gs.alive[nj] = true;
#pragma unroll
                for (int k = 0; k < SEG; ++k)
                    if (wn / 32 + nj == k) {
                        const int sg = en.e[k] & 0xFFFFFF;
                        gs.alive[nj] = sg != 0xFFFFFF;
                        qn = gs.alive[nj] ? sg : s0;
                    }
                const int q1 = qn + l31;
                const bool valid = gs.alive[nj] && q1 < q_end;
                const int qc = valid ? q1 : qn;
                gs.mk[nj][0] = valid ? P.gram_mask0[qc] : 0.f;
                gs.mk[nj][1] = (valid && P.gram_mask1) ? P.gram_mask1[qc] : 0.f;
                gs.anyk[nj][0] = __ballot(gs.mk[nj][0] != 0.f) != 0ull;
                gs.anyk[nj][1] = __ballot(gs.mk[nj][1] != 0.f) != 0ull;
            }
        }
        return gs;
    };
    f32x4 rg[GRAM ? 4 : 1];
    auto issue_g = [&](const GStage& gs, int g) {
        if constexpr (GRAM) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                rg[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gs.f_rsrc, gs.src, (g * 32 + c) * gs.cstride, 0));
        }
    };
    auto convert_store_g = [&](const GStage& gs, int g, int buf) {
        if constexpr (GRAM) {
            typedef _Float16 f16x4_ __attribute__((ext_vector_type(4)));
            const GTask gk = gram_task();
            const int g_half = gk.half, g_kg = gk.kg, g_pos = gk.pos;
            // unit (part, k-group, position) of the buffer's [2][4][BN] image; this thread's half: 8 bytes of the 16
            char* const d = reinterpret_cast<char*>(Rs + buf * HB) + (g_kg * BN + g_pos) * 16 + g_half * 8;
            unsigned bytes = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f16x4_ vh, vl;
                unsigned bits = 0u;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float x = rg[c][j];
                    const float xs = x * gs.fscale;                             // (conv_gram_split's arithmetic)
                    const _Float16 h_ = (_Float16)xs;
                    vh[c] = h_;
                    vl[c] = (_Float16)(xs - (float)h_);
                    bits |= (x > 0.f ? 1u : 0u) << c;
                }
                *reinterpret_cast<f16x4_*>(d + j * 16) = vh;
                *reinterpret_cast<f16x4_*>(d + (4 * BN + j) * 16) = vl;
                bytes |= bits << (8 * j);
            }
            // the partner lane's nibbles -> one byte per position: bit c = channel 8 (4 g + k-group) + c
            const unsigned other = (unsigned)__shfl_xor((int)bytes, 1, 64);
            if (g_half == 0) *reinterpret_cast<unsigned*>(Gb + (g * 4 + g_kg) * BN + g_pos) = bytes | (other << 4);
        }
    };
    // derivative-matrix fragments of Gram stage gs = (unit, mask, k-step of 16 channels) through the weight ring
#define SM_LOAD_D(gs_, slot_)                                                                            \
    {                                                                                                    \
        const int t_ = ((gs_) >> 2) * 2 + ((gs_) & 1), k_ = ((gs_) >> 1) & 1;                            \
        _Pragma("unroll") for (int s = 0; s < NP; ++s)                                                   \
            ra[slot_][s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(              \
                gst.d_rsrc, d_voff, ((t_ * 4 + s * 2) * BM + k_ * (6 * BM * BM / 16)) * 16, 0));         \
    }

    // ---- prologue: the block's first tile, its unit 0 staged; the claim of its second tile
    Entry cur = load_entry(start_x + ((int)blockIdx.x >> 3));
    Stage sp = make_stage(cur);
    issue(sp, 0);
    if (tid == 0) claimed = atomicAdd(counter, 1u);
#pragma unroll
    for (int t = 0; t < AD; ++t) SM_LOAD_A(t, 0);
    convert_store(sp, 0);
    __syncthreads();
    int par = 0;
    float vmax = 0.f;
    const f32x4* const b_frag = Rs + lhi * RP + (wn / 32) * SEGP + l31;         // n-tile i of the wave = window row wn / 32 + i + ky

    for (;;) {
        f32x16 acc[1][NJ];
        f32x16 accg[GRAM ? NJ : 1];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
        int nxt = -1;
        Entry nen = cur;
        GStage gst = make_gstage(cur);
        // unit 1 of this tile: issued AFTER the previous tile's epilogue (registers the epilogue has no room for)
        issue(sp, 1);
        for (int h = 0; h < nh; ++h) {
            {
                // ---- 18 stages on the published buffer
                const f32x4* const bb = b_frag + par * HB;
                f32x4 fb[NJ][NP], fb_next[NJ][NP];
#pragma unroll
                for (int s = 0; s < NP; ++s)
#pragma unroll
                    for (int i = 0; i < NJ; ++i) fb[i][s] = bb[s * 2 * RP + i * SEGP];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int ch = h * 2 + cc;
                    const int ch_next = ch + 1 < n_chunks ? ch + 1 : 0;         // (wraps into the next tile: the same weights)
                    const f32x4* const bc = bb + cc * 4 * RP;
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        // the next stage's fragments are read under this stage's MFMAs (the unit's last stage re-reads its own)
                        const f32x4* const bf_ = tap < 8 ? bc + ((tap + 1) / 3) * SEGP + (tap + 1) % 3 : (cc == 0 ? bb + 4 * RP : bc);
#pragma unroll
                        for (int s = 0; s < NP; ++s)
#pragma unroll
                            for (int i = 0; i < NJ; ++i) fb_next[i][s] = bf_[s * 2 * RP + i * SEGP];
                        f32x4 fa[NP];
#pragma unroll
                        for (int s = 0; s < NP; ++s) fa[s] = ra[tap % AD][s];
#define SM_PIPE_PRODUCT(pa_, pb_)                                                                        \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                       \
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[pa_]),           \
                                                           __builtin_bit_cast(f16x8, fb[j][pb_]), acc[0][j], 0, 0, 0);
                        SM_PIPE_PRODUCT(1, 0)
                        SM_PIPE_PRODUCT(0, 1)
                        SM_PIPE_PRODUCT(0, 0)
#undef SM_PIPE_PRODUCT
                        __builtin_amdgcn_sched_barrier(0);
                        if (tap + AD < 9) {
                            SM_LOAD_A(tap + AD, ch);
                        } else if (GRAM && cc == 1 && h == nh - 1) {            // behind the last window unit: the first Gram stages
                            SM_LOAD_D(tap + AD - 9, tap + AD - 9)
                        } else {
                            SM_LOAD_A(tap + AD - 9, ch_next);
                        }
#pragma unroll
                        for (int s = 0; s < NP; ++s)
#pragma unroll
                            for (int i = 0; i < NJ; ++i) fb[i][s] = fb_next[i][s];
                    }
                }
            }
            // ---- the unit in flight -> the other buffer (its readers passed the previous barrier)
            if (GRAM && h + 1 == nh) {
                convert_store_g(gst, 0, par ^ 1);
            } else {
                convert_store(sp, par ^ 1);
            }
            if (!GRAM && h == nh - 2 && tid == 0) {
                const int sq = g8 + (int)claimed;
                *slot = sq < cnt_x ? start_x + sq : -1;
            }
            __syncthreads();
            par ^= 1;
            // ---- issue the unit after it: (this tile, h + 2), a Gram unit, (next tile, 0) - (next tile, 1): see above
            if (!GRAM && h == nh - 2) {
                nxt = __builtin_amdgcn_readfirstlane(*slot);
                nen = load_entry(nxt >= 0 ? nxt : 0);
                if (nxt >= 0) sp = make_stage(nen);                             // (no next tile: this tile's unit 0 again, unused)
                issue(sp, 0);
            } else if (h + 2 < nh) {
                issue(sp, h + 2);
            } else if (GRAM) {
                issue_g(gst, h + 2 - nh);                                       // (h = nh - 2: Gram unit 0; h = nh - 1: unit 1)
            }
        }
        if constexpr (GRAM) {
            // ---- the two Gram units: masks x k-steps of 32 channels each, in gram_backward_body's order (the sums have its bits)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) accg[j][r] = 0.f;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#define SM_GRAM_STAGE(gs_)                                                                               \
    {                                                                                                    \
        constexpr int k_ = ((gs_) >> 1) & 1, ks_ = (gs_) & 1;                                            \
        f32x4 fa[2];                                                                                     \
        fa[0] = ra[(gs_) % AD][0];                                                                       \
        fa[1] = ra[(gs_) % AD][1];                                                                       \
        _Pragma("unroll") for (int nj = 0; nj < NJ; ++nj) {                                              \
            if (!gst.alive[nj] || !gst.anyk[nj][k_]) continue;   /* (wave-uniform) */                   \
            const f32x4* gf = Gs + (ks_ * 2 + lhi) * BN + nj * 32;                                       \
            const bool keep = gst.mk[nj][k_] != 0.f;                                                     \
            f32x4 fb[2];                                                                                 \
            fb[0] = keep ? gf[0] : zero4;                                                                \
            fb[1] = keep ? gf[4 * BN] : zero4;                                                           \
            conv_gram_mfma(accg[nj], fa, fb);                                                            \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    }
            // stage gs of the tile's eight uses ring slot gs % 3 and refills it with stage gs + 3
            {
                const f32x4* const Gs = Rs + par * HB + wn + l31;
                SM_GRAM_STAGE(0) SM_LOAD_D(3, 0)
                SM_GRAM_STAGE(1) SM_LOAD_D(4, 1)
                SM_GRAM_STAGE(2) SM_LOAD_D(5, 2)
                SM_GRAM_STAGE(3) SM_LOAD_D(6, 0)
            }
            convert_store_g(gst, 1, par ^ 1);
            if (tid == 0) {
                const int sq = g8 + (int)claimed;
                *slot = sq < cnt_x ? start_x + sq : -1;
            }
            __syncthreads();
            par ^= 1;
            nxt = __builtin_amdgcn_readfirstlane(*slot);
            nen = load_entry(nxt >= 0 ? nxt : 0);
            {
                const f32x4* const Gs = Rs + par * HB + wn + l31;
                SM_GRAM_STAGE(4) SM_LOAD_D(7, 1)
                SM_GRAM_STAGE(5)
                SM_GRAM_STAGE(6)
                SM_GRAM_STAGE(7)
            }
#undef SM_GRAM_STAGE
            // (the next tile's first unit travels under the stores: the Gram stages have no registers to spare)
            if (nxt >= 0) sp = make_stage(nen);
            issue(sp, 0);
        }
        // ---- epilogue of the tile; the claim of the tile after next travels under it
        if (tid == 0) claimed = atomicAdd(counter, 1u);
        {
            int qs[SEG];
            bool live[SEG];
            const int s0 = cur.e[0] & 0xFFFFFF;
#pragma unroll
            for (int i = 0; i < SEG; ++i) {
                const int sg = cur.e[i] & 0xFFFFFF;
                live[i] = sg != 0xFFFFFF;
                qs[i] = live[i] ? sg : s0;
            }
            const ConvProblem P = problem_of(cur);
            if constexpr (GRAM) {
                const f32x4 no_bias[1][4] = {};
                vmax = fmaxf(vmax, conv_split_store_tile<BM, BN, WGM, WGN, FLAGS, true>(P, qs, live, acc, 0, out_scale, no_bias, accg,
                                                                                        gst.oscale, Gb));
            } else {
                vmax = fmaxf(vmax, conv_split_epilogue<BM, BN, WGM, WGN, FLAGS, true>(a, P, qs, live, acc, 0, out_scale, bias_s, nullptr));
            }
        }
        if constexpr (GRAM) {
            // (the next tile's first weight stages: behind the stores, whose registers the ring would take)
#pragma unroll
            for (int t = 0; t < AD; ++t) SM_LOAD_A(t, 0);
            convert_store(sp, par ^ 1);
            __syncthreads();
            par ^= 1;
        }
        if (nxt < 0) break;
        cur = nen;
    }
#undef SM_LOAD_D
#undef SM_LOAD_A
    record_amax(a.amax_out, vmax, amax_seen);
}

}  // namespace sm
