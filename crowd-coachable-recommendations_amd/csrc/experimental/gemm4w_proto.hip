// EXPERIMENT (timing only, results are garbage): the four-wave form of the main pass planned in DESIGN.md section 8 item 1.
// One wave per SIMD, 256 threads, the same 256x256 tile / LDS ring / work items as gemm_topk_kernel; each wave owns
// 128 corpus rows x 128 queries = 4 x 4 blocks of v_mfma_f32_32x32x16_bf16 in a[0:255] (asm-owned: every MFMA statement
// lists its accumulator registers as clobbers), the operand fragments of sub-stage u+1 are read between the MFMAs of u,
// the DMA pieces of u+3 between the later ones.  No filter, no output: it answers one question -- how fast the skeleton runs.
// Built only by tools/proto4w.sh into its own shared object; never part of libccr_hip.so.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../ccr_gemm_common.h"

namespace ccr {

#define MFMA_ACC_0(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" ::"v"(A), "v"(B) : "memory", "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15")
#define MFMA_ZERO_0(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15")
#define MFMA_ACC_1(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" ::"v"(A), "v"(B) : "memory", "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31")
#define MFMA_ZERO_1(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31")
#define MFMA_ACC_2(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" ::"v"(A), "v"(B) : "memory", "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47")
#define MFMA_ZERO_2(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47")
#define MFMA_ACC_3(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" ::"v"(A), "v"(B) : "memory", "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63")
#define MFMA_ZERO_3(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63")
#define MFMA_ACC_4(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[64:79], %0, %1, a[64:79]" ::"v"(A), "v"(B) : "memory", "a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79")
#define MFMA_ZERO_4(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[64:79], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79")
#define MFMA_ACC_5(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[80:95], %0, %1, a[80:95]" ::"v"(A), "v"(B) : "memory", "a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95")
#define MFMA_ZERO_5(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[80:95], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95")
#define MFMA_ACC_6(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[96:111], %0, %1, a[96:111]" ::"v"(A), "v"(B) : "memory", "a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111")
#define MFMA_ZERO_6(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[96:111], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111")
#define MFMA_ACC_7(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[112:127], %0, %1, a[112:127]" ::"v"(A), "v"(B) : "memory", "a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127")
#define MFMA_ZERO_7(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[112:127], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127")
#define MFMA_ACC_8(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[128:143], %0, %1, a[128:143]" ::"v"(A), "v"(B) : "memory", "a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143")
#define MFMA_ZERO_8(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[128:143], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143")
#define MFMA_ACC_9(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[144:159], %0, %1, a[144:159]" ::"v"(A), "v"(B) : "memory", "a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159")
#define MFMA_ZERO_9(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[144:159], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159")
#define MFMA_ACC_10(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[160:175], %0, %1, a[160:175]" ::"v"(A), "v"(B) : "memory", "a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175")
#define MFMA_ZERO_10(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[160:175], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175")
#define MFMA_ACC_11(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[176:191], %0, %1, a[176:191]" ::"v"(A), "v"(B) : "memory", "a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191")
#define MFMA_ZERO_11(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[176:191], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191")
#define MFMA_ACC_12(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[192:207], %0, %1, a[192:207]" ::"v"(A), "v"(B) : "memory", "a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207")
#define MFMA_ZERO_12(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[192:207], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207")
#define MFMA_ACC_13(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[208:223], %0, %1, a[208:223]" ::"v"(A), "v"(B) : "memory", "a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223")
#define MFMA_ZERO_13(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[208:223], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223")
#define MFMA_ACC_14(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, a[224:239]" ::"v"(A), "v"(B) : "memory", "a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239")
#define MFMA_ZERO_14(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239")
#define MFMA_ACC_15(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, a[240:255]" ::"v"(A), "v"(B) : "memory", "a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255")
#define MFMA_ZERO_15(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255")
#define DS_READ(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(DST) : "v"(ADDR))
#define WAIT_FRAGS(X) \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]), \
                 "+v"(X[8]), "+v"(X[9]), "+v"(X[10]), "+v"(X[11]), "+v"(X[12]), "+v"(X[13]), "+v"(X[14]), "+v"(X[15])::"memory")

constexpr int P4_THREADS = 256;
#ifndef PROTO_MODE
#define PROTO_MODE 0   // 0: full skeleton; 1: DMA only (64-B row pieces); 2: DMA only, pieces of 8 rows x 128 B (same bytes)
#endif

__global__ __launch_bounds__(P4_THREADS) void gemm4w_proto_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3
    const int wd = wv >> 1, wq = wv & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int KS2 = a.dim / SUB_K;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

    // DMA role: 8 pieces of 1 KiB per sub-stage and wave: rows i*64 + wv*16 + (lane>>2), i = 0..3, of the corpus and of the query half
    const int srow = wv * 16 + (lane >> 2);
    const int schunk = (lane & 3) ^ ((srow >> 2) & 3);
    const int swz = (lane >> 2) & 3;
    const uint32_t a_row = (uint32_t)((wd * 128 + l31) * 64);
    const uint32_t b_row = (uint32_t)(SUB_Q_REGION + (wq * 128 + l31) * 64);
    uint32_t cofs[2];
    for (int ks = 0; ks < 2; ++ks) cofs[ks] = (uint32_t)(((2 * ks + h) ^ swz) << 4);

    const int xcd = blockIdx.x & (NUM_XCD - 1);
    const int jx = blockIdx.x >> 3;
    const int per_x = gridDim.x >> 3;
    const int qg = xcd % a.qgroups;
    const int rc = xcd / a.qgroups;
    const int nrc = NUM_XCD / a.qgroups;
    const int qb_per = a.qblocks / a.qgroups;
    const int rl0 = a.range_begin / nrc;
    const int rl_x = (a.range_end - a.range_begin) / nrc;
    const int count_x = rl_x * qb_per;

    for (int item = jx; item < count_x; item += per_x) {
        const int rl = rl0 + item / qb_per;
        const int qb = qg * qb_per + item % qb_per;
        const int r = rc + nrc * rl;
        const int64_t ntile = (a.n_vt - r + a.ranges - 1) / a.ranges;
        if (ntile <= 0) continue;
        const int q0 = qb * TILE_Q;
        const uint16_t *qsrc[4];
        for (int i = 0; i < 4; ++i) {
            int qrow = q0 + i * 64 + srow;
            if (qrow > a.n_q - 1) qrow = a.n_q - 1;
            qsrc[i] = a.Q + (int64_t)qrow * a.dim + schunk * 8;
        }
        const int64_t U = ntile * KS2;
        int64_t iu = 0, it = 0;
        int iks = 0;
        const uint16_t *dsrc[4];
        auto tile_ptrs = [&]() {
            const int64_t row0 = (r + it * a.ranges) * a.tile_stride * TILE_DOCS;
            for (int i = 0; i < 4; ++i) {
                int64_t drow = row0 + i * 64 + srow;
                if (drow > a.n_rows - 1) drow = a.n_rows - 1;
                dsrc[i] = a.D + drow * a.dim + schunk * 8;
            }
        };
        tile_ptrs();
        // piece p of the sub-stage being issued: 0..3 corpus rows, 4..7 query rows
        auto issue_piece = [&](int p) {
            char *buf = smem + (int)(iu & (RING - 1)) * SUB_BYTES;
            const int k0 = iks * SUB_K;
            if (PROTO_MODE == 2) {
                // even sub-stages: 256 corpus rows x 128 B, odd: 256 query rows x 128 B; piece p = rows p*32 + wv*8 + (lane>>3)
                const int rowp = p * 32 + wv * 8 + (lane >> 3);
                const int kk = (iks >> 1) * 64 + (lane & 7) * 8;
                const uint16_t *src;
                if ((iks & 1) == 0) {
                    int64_t drow = (r + it * a.ranges) * a.tile_stride * TILE_DOCS + rowp;
                    if (drow > a.n_rows - 1) drow = a.n_rows - 1;
                    src = a.D + drow * a.dim + kk;
                } else {
                    int qrow = q0 + rowp;
                    if (qrow > a.n_q - 1) qrow = a.n_q - 1;
                    src = a.Q + (int64_t)qrow * a.dim + kk;
                }
                glds16(src, buf + (p * 256 + wv * 64) * 16);
            } else if (p < 4)
                glds16(dsrc[p] + k0, buf + (p * 256 + wv * 64) * 16);
            else
                glds16(qsrc[p - 4] + k0, buf + SUB_Q_REGION + ((p - 4) * 256 + wv * 64) * 16);
        };
        auto issue_done = [&]() {
            ++iu;
            if (++iks == KS2) {
                iks = 0;
                ++it;
                tile_ptrs();
            }
        };
        auto issue_all = [&]() {
            for (int p = 0; p < 8; ++p) issue_piece(p);
            issue_done();
        };

        const int npro = U < 3 ? (int)U : 3;
        for (int i = 0; i < npro; ++i) issue_all();
        if (npro == 3)
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (npro == 2)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else
            CCR_WAIT_VM(0);
        CCR_BARRIER();

        bf16x8 fa[16], fb[16];   // [0..7] corpus fragments (dt*2+ks), [8..15] query fragments (8+qt*2+ks)
        {
            const uint32_t ra[2] = {lds0 + a_row + cofs[0], lds0 + a_row + cofs[1]};
            const uint32_t rb[2] = {lds0 + b_row + cofs[0], lds0 + b_row + cofs[1]};
            bf16x8(&nxt)[16] = fa;
            DS_READ(nxt[0], ra[0], 0); DS_READ(nxt[1], ra[1], 0); DS_READ(nxt[2], ra[0], 2048); DS_READ(nxt[3], ra[1], 2048);
            DS_READ(nxt[4], ra[0], 4096); DS_READ(nxt[5], ra[1], 4096); DS_READ(nxt[6], ra[0], 6144); DS_READ(nxt[7], ra[1], 6144);
            DS_READ(nxt[8], rb[0], 0); DS_READ(nxt[9], rb[1], 0); DS_READ(nxt[10], rb[0], 2048); DS_READ(nxt[11], rb[1], 2048);
            DS_READ(nxt[12], rb[0], 4096); DS_READ(nxt[13], rb[1], 4096); DS_READ(nxt[14], rb[0], 6144); DS_READ(nxt[15], rb[1], 6144);
            WAIT_FRAGS(fa);
        }

        int cks = 0;
        auto step = [&](bf16x8(&cur)[16], bf16x8(&nxt)[16], int64_t u) {
            // own pieces of u+1 landed (those of u+2 may still fly), then everybody's
            if (u + 1 < U) {
                if (u + 2 < U)
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else
                    CCR_WAIT_VM(0);
            }
            CCR_BARRIER();
            const uint32_t nb = lds0 + (uint32_t)((u + 1) & (RING - 1)) * SUB_BYTES;
            const uint32_t ra[2] = {nb + a_row + cofs[0], nb + a_row + cofs[1]};
            const uint32_t rb[2] = {nb + b_row + cofs[0], nb + b_row + cofs[1]};
            const bool more = u + 3 < U;
            if (PROTO_MODE != 0) {
                if (more) for (int p = 0; p < 8; ++p) issue_piece(p);
            } else if (cks == 0) {
        MFMA_ZERO_0(cur[0], cur[8]);
        DS_READ(nxt[0], ra[0], 0);
        MFMA_ZERO_1(cur[0], cur[10]);
        MFMA_ZERO_2(cur[0], cur[12]);
        DS_READ(nxt[1], ra[1], 0);
        MFMA_ZERO_3(cur[0], cur[14]);
        if (more) issue_piece(0);
        MFMA_ZERO_4(cur[2], cur[8]);
        DS_READ(nxt[2], ra[0], 2048);
        MFMA_ZERO_5(cur[2], cur[10]);
        MFMA_ZERO_6(cur[2], cur[12]);
        DS_READ(nxt[3], ra[1], 2048);
        MFMA_ZERO_7(cur[2], cur[14]);
        if (more) issue_piece(1);
        MFMA_ZERO_8(cur[4], cur[8]);
        DS_READ(nxt[4], ra[0], 4096);
        MFMA_ZERO_9(cur[4], cur[10]);
        MFMA_ZERO_10(cur[4], cur[12]);
        DS_READ(nxt[5], ra[1], 4096);
        MFMA_ZERO_11(cur[4], cur[14]);
        if (more) issue_piece(2);
        MFMA_ZERO_12(cur[6], cur[8]);
        DS_READ(nxt[6], ra[0], 6144);
        MFMA_ZERO_13(cur[6], cur[10]);
        MFMA_ZERO_14(cur[6], cur[12]);
        DS_READ(nxt[7], ra[1], 6144);
        MFMA_ZERO_15(cur[6], cur[14]);
        if (more) issue_piece(3);
        MFMA_ACC_0(cur[1], cur[9]);
        DS_READ(nxt[8], rb[0], 0);
        MFMA_ACC_1(cur[1], cur[11]);
        MFMA_ACC_2(cur[1], cur[13]);
        DS_READ(nxt[9], rb[1], 0);
        MFMA_ACC_3(cur[1], cur[15]);
        if (more) issue_piece(4);
        MFMA_ACC_4(cur[3], cur[9]);
        DS_READ(nxt[10], rb[0], 2048);
        MFMA_ACC_5(cur[3], cur[11]);
        MFMA_ACC_6(cur[3], cur[13]);
        DS_READ(nxt[11], rb[1], 2048);
        MFMA_ACC_7(cur[3], cur[15]);
        if (more) issue_piece(5);
        MFMA_ACC_8(cur[5], cur[9]);
        DS_READ(nxt[12], rb[0], 4096);
        MFMA_ACC_9(cur[5], cur[11]);
        MFMA_ACC_10(cur[5], cur[13]);
        DS_READ(nxt[13], rb[1], 4096);
        MFMA_ACC_11(cur[5], cur[15]);
        if (more) issue_piece(6);
        MFMA_ACC_12(cur[7], cur[9]);
        DS_READ(nxt[14], rb[0], 6144);
        MFMA_ACC_13(cur[7], cur[11]);
        MFMA_ACC_14(cur[7], cur[13]);
        DS_READ(nxt[15], rb[1], 6144);
        MFMA_ACC_15(cur[7], cur[15]);
        if (more) issue_piece(7);
            } else {
        MFMA_ACC_0(cur[0], cur[8]);
        DS_READ(nxt[0], ra[0], 0);
        MFMA_ACC_1(cur[0], cur[10]);
        MFMA_ACC_2(cur[0], cur[12]);
        DS_READ(nxt[1], ra[1], 0);
        MFMA_ACC_3(cur[0], cur[14]);
        if (more) issue_piece(0);
        MFMA_ACC_4(cur[2], cur[8]);
        DS_READ(nxt[2], ra[0], 2048);
        MFMA_ACC_5(cur[2], cur[10]);
        MFMA_ACC_6(cur[2], cur[12]);
        DS_READ(nxt[3], ra[1], 2048);
        MFMA_ACC_7(cur[2], cur[14]);
        if (more) issue_piece(1);
        MFMA_ACC_8(cur[4], cur[8]);
        DS_READ(nxt[4], ra[0], 4096);
        MFMA_ACC_9(cur[4], cur[10]);
        MFMA_ACC_10(cur[4], cur[12]);
        DS_READ(nxt[5], ra[1], 4096);
        MFMA_ACC_11(cur[4], cur[14]);
        if (more) issue_piece(2);
        MFMA_ACC_12(cur[6], cur[8]);
        DS_READ(nxt[6], ra[0], 6144);
        MFMA_ACC_13(cur[6], cur[10]);
        MFMA_ACC_14(cur[6], cur[12]);
        DS_READ(nxt[7], ra[1], 6144);
        MFMA_ACC_15(cur[6], cur[14]);
        if (more) issue_piece(3);
        MFMA_ACC_0(cur[1], cur[9]);
        DS_READ(nxt[8], rb[0], 0);
        MFMA_ACC_1(cur[1], cur[11]);
        MFMA_ACC_2(cur[1], cur[13]);
        DS_READ(nxt[9], rb[1], 0);
        MFMA_ACC_3(cur[1], cur[15]);
        if (more) issue_piece(4);
        MFMA_ACC_4(cur[3], cur[9]);
        DS_READ(nxt[10], rb[0], 2048);
        MFMA_ACC_5(cur[3], cur[11]);
        MFMA_ACC_6(cur[3], cur[13]);
        DS_READ(nxt[11], rb[1], 2048);
        MFMA_ACC_7(cur[3], cur[15]);
        if (more) issue_piece(5);
        MFMA_ACC_8(cur[5], cur[9]);
        DS_READ(nxt[12], rb[0], 4096);
        MFMA_ACC_9(cur[5], cur[11]);
        MFMA_ACC_10(cur[5], cur[13]);
        DS_READ(nxt[13], rb[1], 4096);
        MFMA_ACC_11(cur[5], cur[15]);
        if (more) issue_piece(6);
        MFMA_ACC_12(cur[7], cur[9]);
        DS_READ(nxt[14], rb[0], 6144);
        MFMA_ACC_13(cur[7], cur[11]);
        MFMA_ACC_14(cur[7], cur[13]);
        DS_READ(nxt[15], rb[1], 6144);
        MFMA_ACC_15(cur[7], cur[15]);
        if (more) issue_piece(7);
            }
            if (more) issue_done();
            if (PROTO_MODE == 0) WAIT_FRAGS(nxt);
            if (++cks == KS2) cks = 0;
        };
        for (int64_t u = 0; u < U; u += 2) {
            step(fa, fb, u);
            if (u + 1 < U) step(fb, fa, u + 1);
        }
        __syncthreads();
    }
}

}  // namespace ccr

// exported for tools/proto4w.py: times `reps` launches over an NQ-shaped problem with the production work-item mapping
extern "C" int ccr_proto4w_time(const uint16_t *D, int64_t n_rows, int dim, const uint16_t *Q, int n_q, int ranges, int qgroups,
                                int reps, float *ms_out) {
    using namespace ccr;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.D = D; g.n_rows = n_rows; g.dim = dim; g.Q = Q; g.n_q = n_q;
    g.nq_pad = (n_q + TILE_Q - 1) / TILE_Q * TILE_Q;
    g.qblocks = g.nq_pad / TILE_Q;
    g.n_vt = (n_rows + TILE_DOCS - 1) / TILE_DOCS;
    g.tile_stride = 1;
    g.ranges = ranges;
    g.qgroups = qgroups;
    g.range_begin = 0;
    g.range_end = ranges;
    const size_t lds = RING * (size_t)SUB_BYTES;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm4w_proto_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(gemm4w_proto_kernel, dim3(256), dim3(P4_THREADS), lds, 0, g);   // warm-up
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm4w_proto_kernel, dim3(256), dim3(P4_THREADS), lds, 0, g);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return -2;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
