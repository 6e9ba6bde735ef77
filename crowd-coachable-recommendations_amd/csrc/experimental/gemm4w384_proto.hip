// EXPERIMENT (timing only, results are garbage): four waves, 256 x 384 tile -- the "fewer fill bytes per flop" form of DESIGN.md
// section 8 item 1.  Each wave owns 128 corpus rows x 192 queries = 4 x 6 blocks of v_mfma_f32_32x32x16_bf16: 16 blocks in
// a[0:255] (asm-owned), 8 blocks in compiler-allocated VGPRs ("+v" operands of the asm MFMAs).  Sub-stage = 40 KiB
// (256 corpus rows + 384 query rows x 64 B), ring of three; one barrier per sub-stage in the middle of it:
//   K0(u): MFMAs of K half 0  ||  reads of the K-half-1 fragments of buffer u
//   wait own DMA of u+1, barrier
//   K1(u): MFMAs of K half 1  ||  DMA pieces of u+3 into buffer u  ||  reads of the K-half-0 fragments of buffer u+1
// n_rows must be a multiple of 256 and n_q of 384 (no tail clamps).  Built only by tools/proto4w.sh.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../ccr_gemm_common.h"

namespace ccr {

#define MA_0(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" ::"v"(A), "v"(B) : "memory", "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15")
#define MZ_0(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15")
#define MA_1(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" ::"v"(A), "v"(B) : "memory", "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31")
#define MZ_1(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31")
#define MA_2(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" ::"v"(A), "v"(B) : "memory", "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47")
#define MZ_2(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47")
#define MA_3(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" ::"v"(A), "v"(B) : "memory", "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63")
#define MZ_3(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63")
#define MA_4(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[64:79], %0, %1, a[64:79]" ::"v"(A), "v"(B) : "memory", "a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79")
#define MZ_4(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[64:79], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79")
#define MA_5(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[80:95], %0, %1, a[80:95]" ::"v"(A), "v"(B) : "memory", "a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95")
#define MZ_5(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[80:95], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95")
#define MA_6(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[96:111], %0, %1, a[96:111]" ::"v"(A), "v"(B) : "memory", "a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111")
#define MZ_6(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[96:111], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111")
#define MA_7(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[112:127], %0, %1, a[112:127]" ::"v"(A), "v"(B) : "memory", "a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127")
#define MZ_7(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[112:127], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127")
#define MA_8(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[128:143], %0, %1, a[128:143]" ::"v"(A), "v"(B) : "memory", "a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143")
#define MZ_8(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[128:143], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143")
#define MA_9(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[144:159], %0, %1, a[144:159]" ::"v"(A), "v"(B) : "memory", "a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159")
#define MZ_9(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[144:159], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159")
#define MA_10(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[160:175], %0, %1, a[160:175]" ::"v"(A), "v"(B) : "memory", "a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175")
#define MZ_10(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[160:175], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175")
#define MA_11(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[176:191], %0, %1, a[176:191]" ::"v"(A), "v"(B) : "memory", "a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191")
#define MZ_11(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[176:191], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191")
#define MA_12(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[192:207], %0, %1, a[192:207]" ::"v"(A), "v"(B) : "memory", "a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207")
#define MZ_12(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[192:207], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207")
#define MA_13(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[208:223], %0, %1, a[208:223]" ::"v"(A), "v"(B) : "memory", "a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223")
#define MZ_13(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[208:223], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223")
#define MA_14(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, a[224:239]" ::"v"(A), "v"(B) : "memory", "a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239")
#define MZ_14(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[224:239], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239")
#define MA_15(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, a[240:255]" ::"v"(A), "v"(B) : "memory", "a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255")
#define MZ_15(A, B) asm volatile("v_mfma_f32_32x32x16_bf16 a[240:255], %0, %1, 0" ::"v"(A), "v"(B) : "memory", "a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255")
#define DS_READ(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(DST) : "v"(ADDR))
#define WAIT10(A, B) \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]), \
                 "+v"(B[4]), "+v"(B[5])::"memory")

constexpr int P4_THREADS = 256;
#ifndef PROTO_MODE
#define PROTO_MODE 0   // 0: timing skeleton; 3: test build that stores every raw MFMA score (a.store [n_q][n_rows])
#endif
constexpr int TQ = 384;
constexpr int SUBB = (256 + TQ) * 64;   // 40960
constexpr int QOFF = 256 * 64;          // 16384
constexpr int RING3 = 3;

__global__ __launch_bounds__(P4_THREADS) void gemm4w384_proto_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wd = wv >> 1, wq = wv & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int KS2 = a.dim / SUB_K;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int srow = wv * 16 + (lane >> 2);
    const int schunk = (lane & 3) ^ ((srow >> 2) & 3);
    const int swz = (lane >> 2) & 3;
    const uint32_t a_row = (uint32_t)((wd * 128 + l31) * 64);
    const uint32_t b_row = (uint32_t)(QOFF + (wq * 192 + l31) * 64);
    const uint32_t cofs0 = (uint32_t)(((0 + h) ^ swz) << 4), cofs1 = (uint32_t)(((2 + h) ^ swz) << 4);
    const int64_t piece_stride = (int64_t)64 * a.dim;   // elements between the rows of consecutive pieces

    const int xcd = blockIdx.x & (NUM_XCD - 1);
    const int jx = blockIdx.x >> 3;
    const int per_x = gridDim.x >> 3;
    const int qblocks = a.n_q / TQ;
    const int rl_x = a.ranges / NUM_XCD;
    const int count_x = rl_x * qblocks;

    for (int item = jx; item < count_x; item += per_x) {
        const int rl = item / qblocks;
        const int qb = item % qblocks;
        const int r = xcd + NUM_XCD * rl;
        const int64_t ntile = (a.n_vt - r + a.ranges - 1) / a.ranges;
        if (ntile <= 0) continue;
        const uint16_t *qsrc = a.Q + (int64_t)(qb * TQ + srow) * a.dim + schunk * 8;
        const int64_t U = ntile * KS2;
        int64_t it = 0;
        int iks = 0, ibuf = 0;
        const uint16_t *dsrc = a.D + ((int64_t)r * TILE_DOCS + srow) * a.dim + schunk * 8;
        auto issue_piece = [&](int p) {
            char *buf = smem + ibuf * SUBB;
            const int k0 = iks * SUB_K;
            if (p < 4)
                glds16(dsrc + p * piece_stride + k0, buf + (p * 64 + wv * 16) * 64);
            else
                glds16(qsrc + (p - 4) * piece_stride + k0, buf + QOFF + ((p - 4) * 64 + wv * 16) * 64);
        };
        auto issue_done = [&]() {
            if (++ibuf == RING3) ibuf = 0;
            if (++iks == KS2) {
                iks = 0;
                ++it;
                dsrc = a.D + ((int64_t)(r + it * a.ranges) * TILE_DOCS + srow) * a.dim + schunk * 8;
            }
        };
        const int npro = U < 3 ? (int)U : 3;
        for (int i = 0; i < npro; ++i) {
            for (int p = 0; p < 10; ++p) issue_piece(p);
            issue_done();
        }
        if (npro == 3)
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (npro == 2)
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else
            CCR_WAIT_VM(0);
        CCR_BARRIER();

        bf16x8 A0[4], A1[4], B0[6], B1[6];
        f32x16 accv[8];
        {
            const uint32_t na0 = lds0 + a_row + cofs0, nb0 = lds0 + b_row + cofs0;
            DS_READ(A0[0], na0, 0); DS_READ(A0[1], na0, 2048); DS_READ(A0[2], na0, 4096); DS_READ(A0[3], na0, 6144);
            DS_READ(B0[0], nb0, 0); DS_READ(B0[1], nb0, 2048); DS_READ(B0[2], nb0, 4096); DS_READ(B0[3], nb0, 6144);
            DS_READ(B0[4], nb0, 8192); DS_READ(B0[5], nb0, 10240);
            WAIT10(A0, B0);
        }
        int cks = 0, cbuf = 0;
        int64_t ct = 0;
        (void)ct;
        for (int64_t u = 0; u < U; ++u) {
            const uint32_t cb = lds0 + (uint32_t)cbuf * SUBB;
            const int nbuf = cbuf + 1 == RING3 ? 0 : cbuf + 1;
            const uint32_t nb = lds0 + (uint32_t)nbuf * SUBB;
            const uint32_t ra1 = cb + a_row + cofs1, rb1 = cb + b_row + cofs1;
            const uint32_t na0 = nb + a_row + cofs0, nb0 = nb + b_row + cofs0;
            const bool more = u + 3 < U, have_next = u + 1 < U;
            if (cks == 0) {
        MZ_0(A0[0], B0[0]);
        MZ_1(A0[0], B0[1]);
        DS_READ(A1[0], ra1, 0);
        MZ_2(A0[0], B0[2]);
        MZ_3(A0[0], B0[3]);
        DS_READ(A1[1], ra1, 2048);
        MZ_4(A0[0], B0[4]);
        MZ_5(A0[0], B0[5]);
        DS_READ(A1[2], ra1, 4096);
        MZ_6(A0[1], B0[0]);
        MZ_7(A0[1], B0[1]);
        DS_READ(A1[3], ra1, 6144);
        MZ_8(A0[1], B0[2]);
        MZ_9(A0[1], B0[3]);
        DS_READ(B1[0], rb1, 0);
        MZ_10(A0[1], B0[4]);
        MZ_11(A0[1], B0[5]);
        DS_READ(B1[1], rb1, 2048);
        MZ_12(A0[2], B0[0]);
        MZ_13(A0[2], B0[1]);
        DS_READ(B1[2], rb1, 4096);
        MZ_14(A0[2], B0[2]);
        MZ_15(A0[2], B0[3]);
        DS_READ(B1[3], rb1, 6144);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(accv[0]) : "v"(A0[2]), "v"(B0[4]) : "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(accv[1]) : "v"(A0[2]), "v"(B0[5]) : "memory");
        DS_READ(B1[4], rb1, 8192);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(accv[2]) : "v"(A0[3]), "v"(B0[0]) : "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(accv[3]) : "v"(A0[3]), "v"(B0[1]) : "memory");
        DS_READ(B1[5], rb1, 10240);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(accv[4]) : "v"(A0[3]), "v"(B0[2]) : "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(accv[5]) : "v"(A0[3]), "v"(B0[3]) : "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(accv[6]) : "v"(A0[3]), "v"(B0[4]) : "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(accv[7]) : "v"(A0[3]), "v"(B0[5]) : "memory");
            } else {
        MA_0(A0[0], B0[0]);
        MA_1(A0[0], B0[1]);
        DS_READ(A1[0], ra1, 0);
        MA_2(A0[0], B0[2]);
        MA_3(A0[0], B0[3]);
        DS_READ(A1[1], ra1, 2048);
        MA_4(A0[0], B0[4]);
        MA_5(A0[0], B0[5]);
        DS_READ(A1[2], ra1, 4096);
        MA_6(A0[1], B0[0]);
        MA_7(A0[1], B0[1]);
        DS_READ(A1[3], ra1, 6144);
        MA_8(A0[1], B0[2]);
        MA_9(A0[1], B0[3]);
        DS_READ(B1[0], rb1, 0);
        MA_10(A0[1], B0[4]);
        MA_11(A0[1], B0[5]);
        DS_READ(B1[1], rb1, 2048);
        MA_12(A0[2], B0[0]);
        MA_13(A0[2], B0[1]);
        DS_READ(B1[2], rb1, 4096);
        MA_14(A0[2], B0[2]);
        MA_15(A0[2], B0[3]);
        DS_READ(B1[3], rb1, 6144);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[0]) : "v"(A0[2]), "v"(B0[4]) : "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[1]) : "v"(A0[2]), "v"(B0[5]) : "memory");
        DS_READ(B1[4], rb1, 8192);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[2]) : "v"(A0[3]), "v"(B0[0]) : "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[3]) : "v"(A0[3]), "v"(B0[1]) : "memory");
        DS_READ(B1[5], rb1, 10240);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[4]) : "v"(A0[3]), "v"(B0[2]) : "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[5]) : "v"(A0[3]), "v"(B0[3]) : "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[6]) : "v"(A0[3]), "v"(B0[4]) : "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[7]) : "v"(A0[3]), "v"(B0[5]) : "memory");
            }
            WAIT10(A1, B1);
            if (have_next) {
                if (u + 2 < U)
                    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else
                    CCR_WAIT_VM(0);
            }
            CCR_BARRIER();
        MA_0(A1[0], B1[0]);
        if (more) issue_piece(0);
        MA_1(A1[0], B1[1]);
        if (have_next) DS_READ(A0[0], na0, 0);
        MA_2(A1[0], B1[2]);
        if (more) issue_piece(1);
        MA_3(A1[0], B1[3]);
        if (have_next) DS_READ(A0[1], na0, 2048);
        MA_4(A1[0], B1[4]);
        if (more) issue_piece(2);
        MA_5(A1[0], B1[5]);
        if (have_next) DS_READ(A0[2], na0, 4096);
        MA_6(A1[1], B1[0]);
        if (more) issue_piece(3);
        MA_7(A1[1], B1[1]);
        if (have_next) DS_READ(A0[3], na0, 6144);
        MA_8(A1[1], B1[2]);
        if (more) issue_piece(4);
        MA_9(A1[1], B1[3]);
        if (have_next) DS_READ(B0[0], nb0, 0);
        MA_10(A1[1], B1[4]);
        if (more) issue_piece(5);
        MA_11(A1[1], B1[5]);
        if (have_next) DS_READ(B0[1], nb0, 2048);
        MA_12(A1[2], B1[0]);
        if (more) issue_piece(6);
        MA_13(A1[2], B1[1]);
        if (have_next) DS_READ(B0[2], nb0, 4096);
        MA_14(A1[2], B1[2]);
        if (more) issue_piece(7);
        MA_15(A1[2], B1[3]);
        if (have_next) DS_READ(B0[3], nb0, 6144);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[0]) : "v"(A1[2]), "v"(B1[4]) : "memory");
        if (more) issue_piece(8);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[1]) : "v"(A1[2]), "v"(B1[5]) : "memory");
        if (have_next) DS_READ(B0[4], nb0, 8192);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[2]) : "v"(A1[3]), "v"(B1[0]) : "memory");
        if (more) issue_piece(9);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[3]) : "v"(A1[3]), "v"(B1[1]) : "memory");
        if (have_next) DS_READ(B0[5], nb0, 10240);
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[4]) : "v"(A1[3]), "v"(B1[2]) : "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[5]) : "v"(A1[3]), "v"(B1[3]) : "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[6]) : "v"(A1[3]), "v"(B1[4]) : "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(accv[7]) : "v"(A1[3]), "v"(B1[5]) : "memory");
            if (more) issue_done();
            WAIT10(A0, B0);
            if (PROTO_MODE == 3 && cks == KS2 - 1) {   // test build: write the finished tile's raw MFMA scores
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
                const int64_t row_t = (int64_t)(r + ct * a.ranges) * TILE_DOCS + wd * 128 + 4 * h;
                const int64_t qcol = (int64_t)qb * TQ + wq * 192 + l31;
                { float x; asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 0] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a1" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 1] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a2" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 2] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a3" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 3] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a4" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 8] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a5" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 9] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a6" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 10] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a7" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 11] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a8" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 16] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a9" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 17] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a10" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 18] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a11" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 19] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a12" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 24] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a13" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 25] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a14" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 26] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a15" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 27] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a16" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 0] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a17" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 1] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a18" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 2] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a19" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 3] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a20" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 8] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a21" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 9] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a22" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 10] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a23" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 11] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a24" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 16] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a25" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 17] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a26" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 18] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a27" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 19] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a28" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 24] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a29" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 25] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a30" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 26] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a31" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 27] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a32" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 0] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a33" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 1] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a34" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 2] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a35" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 3] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a36" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 8] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a37" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 9] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a38" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 10] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a39" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 11] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a40" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 16] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a41" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 17] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a42" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 18] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a43" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 19] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a44" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 24] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a45" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 25] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a46" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 26] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a47" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 27] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a48" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 0] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a49" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 1] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a50" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 2] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a51" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 3] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a52" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 8] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a53" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 9] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a54" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 10] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a55" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 11] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a56" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 16] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a57" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 17] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a58" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 18] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a59" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 19] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a60" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 24] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a61" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 25] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a62" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 26] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a63" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 27] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a64" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 0] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a65" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 1] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a66" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 2] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a67" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 3] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a68" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 8] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a69" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 9] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a70" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 10] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a71" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 11] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a72" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 16] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a73" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 17] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a74" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 18] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a75" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 19] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a76" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 24] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a77" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 25] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a78" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 26] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a79" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 27] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a80" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 0] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a81" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 1] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a82" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 2] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a83" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 3] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a84" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 8] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a85" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 9] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a86" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 10] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a87" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 11] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a88" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 16] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a89" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 17] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a90" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 18] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a91" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 19] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a92" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 24] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a93" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 25] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a94" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 26] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a95" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 27] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a96" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 32] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a97" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 33] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a98" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 34] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a99" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 35] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a100" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 40] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a101" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 41] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a102" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 42] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a103" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 43] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a104" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 48] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a105" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 49] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a106" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 50] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a107" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 51] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a108" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 56] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a109" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 57] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a110" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 58] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a111" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 59] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a112" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 32] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a113" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 33] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a114" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 34] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a115" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 35] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a116" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 40] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a117" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 41] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a118" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 42] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a119" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 43] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a120" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 48] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a121" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 49] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a122" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 50] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a123" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 51] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a124" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 56] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a125" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 57] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a126" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 58] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a127" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 59] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a128" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 32] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a129" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 33] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a130" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 34] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a131" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 35] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a132" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 40] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a133" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 41] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a134" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 42] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a135" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 43] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a136" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 48] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a137" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 49] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a138" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 50] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a139" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 51] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a140" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 56] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a141" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 57] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a142" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 58] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a143" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 59] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a144" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 32] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a145" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 33] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a146" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 34] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a147" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 35] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a148" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 40] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a149" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 41] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a150" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 42] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a151" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 43] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a152" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 48] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a153" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 49] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a154" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 50] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a155" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 51] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a156" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 56] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a157" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 57] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a158" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 58] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a159" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 59] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a160" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 32] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a161" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 33] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a162" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 34] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a163" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 35] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a164" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 40] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a165" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 41] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a166" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 42] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a167" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 43] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a168" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 48] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a169" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 49] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a170" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 50] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a171" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 51] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a172" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 56] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a173" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 57] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a174" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 58] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a175" : "=v"(x)); a.store[(qcol + 128) * a.n_rows + row_t + 59] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a176" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 32] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a177" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 33] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a178" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 34] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a179" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 35] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a180" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 40] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a181" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 41] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a182" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 42] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a183" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 43] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a184" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 48] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a185" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 49] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a186" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 50] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a187" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 51] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a188" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 56] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a189" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 57] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a190" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 58] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a191" : "=v"(x)); a.store[(qcol + 160) * a.n_rows + row_t + 59] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a192" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 64] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a193" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 65] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a194" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 66] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a195" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 67] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a196" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 72] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a197" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 73] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a198" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 74] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a199" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 75] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a200" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 80] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a201" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 81] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a202" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 82] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a203" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 83] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a204" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 88] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a205" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 89] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a206" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 90] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a207" : "=v"(x)); a.store[(qcol + 0) * a.n_rows + row_t + 91] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a208" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 64] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a209" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 65] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a210" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 66] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a211" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 67] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a212" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 72] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a213" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 73] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a214" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 74] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a215" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 75] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a216" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 80] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a217" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 81] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a218" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 82] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a219" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 83] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a220" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 88] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a221" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 89] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a222" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 90] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a223" : "=v"(x)); a.store[(qcol + 32) * a.n_rows + row_t + 91] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a224" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 64] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a225" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 65] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a226" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 66] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a227" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 67] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a228" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 72] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a229" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 73] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a230" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 74] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a231" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 75] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a232" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 80] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a233" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 81] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a234" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 82] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a235" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 83] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a236" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 88] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a237" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 89] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a238" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 90] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a239" : "=v"(x)); a.store[(qcol + 64) * a.n_rows + row_t + 91] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a240" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 64] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a241" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 65] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a242" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 66] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a243" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 67] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a244" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 72] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a245" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 73] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a246" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 74] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a247" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 75] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a248" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 80] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a249" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 81] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a250" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 82] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a251" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 83] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a252" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 88] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a253" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 89] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a254" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 90] = x; }
                { float x; asm volatile("v_accvgpr_read_b32 %0, a255" : "=v"(x)); a.store[(qcol + 96) * a.n_rows + row_t + 91] = x; }
                a.store[(qcol + 128) * a.n_rows + row_t + 64] = accv[0][0];
                a.store[(qcol + 128) * a.n_rows + row_t + 65] = accv[0][1];
                a.store[(qcol + 128) * a.n_rows + row_t + 66] = accv[0][2];
                a.store[(qcol + 128) * a.n_rows + row_t + 67] = accv[0][3];
                a.store[(qcol + 128) * a.n_rows + row_t + 72] = accv[0][4];
                a.store[(qcol + 128) * a.n_rows + row_t + 73] = accv[0][5];
                a.store[(qcol + 128) * a.n_rows + row_t + 74] = accv[0][6];
                a.store[(qcol + 128) * a.n_rows + row_t + 75] = accv[0][7];
                a.store[(qcol + 128) * a.n_rows + row_t + 80] = accv[0][8];
                a.store[(qcol + 128) * a.n_rows + row_t + 81] = accv[0][9];
                a.store[(qcol + 128) * a.n_rows + row_t + 82] = accv[0][10];
                a.store[(qcol + 128) * a.n_rows + row_t + 83] = accv[0][11];
                a.store[(qcol + 128) * a.n_rows + row_t + 88] = accv[0][12];
                a.store[(qcol + 128) * a.n_rows + row_t + 89] = accv[0][13];
                a.store[(qcol + 128) * a.n_rows + row_t + 90] = accv[0][14];
                a.store[(qcol + 128) * a.n_rows + row_t + 91] = accv[0][15];
                a.store[(qcol + 160) * a.n_rows + row_t + 64] = accv[1][0];
                a.store[(qcol + 160) * a.n_rows + row_t + 65] = accv[1][1];
                a.store[(qcol + 160) * a.n_rows + row_t + 66] = accv[1][2];
                a.store[(qcol + 160) * a.n_rows + row_t + 67] = accv[1][3];
                a.store[(qcol + 160) * a.n_rows + row_t + 72] = accv[1][4];
                a.store[(qcol + 160) * a.n_rows + row_t + 73] = accv[1][5];
                a.store[(qcol + 160) * a.n_rows + row_t + 74] = accv[1][6];
                a.store[(qcol + 160) * a.n_rows + row_t + 75] = accv[1][7];
                a.store[(qcol + 160) * a.n_rows + row_t + 80] = accv[1][8];
                a.store[(qcol + 160) * a.n_rows + row_t + 81] = accv[1][9];
                a.store[(qcol + 160) * a.n_rows + row_t + 82] = accv[1][10];
                a.store[(qcol + 160) * a.n_rows + row_t + 83] = accv[1][11];
                a.store[(qcol + 160) * a.n_rows + row_t + 88] = accv[1][12];
                a.store[(qcol + 160) * a.n_rows + row_t + 89] = accv[1][13];
                a.store[(qcol + 160) * a.n_rows + row_t + 90] = accv[1][14];
                a.store[(qcol + 160) * a.n_rows + row_t + 91] = accv[1][15];
                a.store[(qcol + 0) * a.n_rows + row_t + 96] = accv[2][0];
                a.store[(qcol + 0) * a.n_rows + row_t + 97] = accv[2][1];
                a.store[(qcol + 0) * a.n_rows + row_t + 98] = accv[2][2];
                a.store[(qcol + 0) * a.n_rows + row_t + 99] = accv[2][3];
                a.store[(qcol + 0) * a.n_rows + row_t + 104] = accv[2][4];
                a.store[(qcol + 0) * a.n_rows + row_t + 105] = accv[2][5];
                a.store[(qcol + 0) * a.n_rows + row_t + 106] = accv[2][6];
                a.store[(qcol + 0) * a.n_rows + row_t + 107] = accv[2][7];
                a.store[(qcol + 0) * a.n_rows + row_t + 112] = accv[2][8];
                a.store[(qcol + 0) * a.n_rows + row_t + 113] = accv[2][9];
                a.store[(qcol + 0) * a.n_rows + row_t + 114] = accv[2][10];
                a.store[(qcol + 0) * a.n_rows + row_t + 115] = accv[2][11];
                a.store[(qcol + 0) * a.n_rows + row_t + 120] = accv[2][12];
                a.store[(qcol + 0) * a.n_rows + row_t + 121] = accv[2][13];
                a.store[(qcol + 0) * a.n_rows + row_t + 122] = accv[2][14];
                a.store[(qcol + 0) * a.n_rows + row_t + 123] = accv[2][15];
                a.store[(qcol + 32) * a.n_rows + row_t + 96] = accv[3][0];
                a.store[(qcol + 32) * a.n_rows + row_t + 97] = accv[3][1];
                a.store[(qcol + 32) * a.n_rows + row_t + 98] = accv[3][2];
                a.store[(qcol + 32) * a.n_rows + row_t + 99] = accv[3][3];
                a.store[(qcol + 32) * a.n_rows + row_t + 104] = accv[3][4];
                a.store[(qcol + 32) * a.n_rows + row_t + 105] = accv[3][5];
                a.store[(qcol + 32) * a.n_rows + row_t + 106] = accv[3][6];
                a.store[(qcol + 32) * a.n_rows + row_t + 107] = accv[3][7];
                a.store[(qcol + 32) * a.n_rows + row_t + 112] = accv[3][8];
                a.store[(qcol + 32) * a.n_rows + row_t + 113] = accv[3][9];
                a.store[(qcol + 32) * a.n_rows + row_t + 114] = accv[3][10];
                a.store[(qcol + 32) * a.n_rows + row_t + 115] = accv[3][11];
                a.store[(qcol + 32) * a.n_rows + row_t + 120] = accv[3][12];
                a.store[(qcol + 32) * a.n_rows + row_t + 121] = accv[3][13];
                a.store[(qcol + 32) * a.n_rows + row_t + 122] = accv[3][14];
                a.store[(qcol + 32) * a.n_rows + row_t + 123] = accv[3][15];
                a.store[(qcol + 64) * a.n_rows + row_t + 96] = accv[4][0];
                a.store[(qcol + 64) * a.n_rows + row_t + 97] = accv[4][1];
                a.store[(qcol + 64) * a.n_rows + row_t + 98] = accv[4][2];
                a.store[(qcol + 64) * a.n_rows + row_t + 99] = accv[4][3];
                a.store[(qcol + 64) * a.n_rows + row_t + 104] = accv[4][4];
                a.store[(qcol + 64) * a.n_rows + row_t + 105] = accv[4][5];
                a.store[(qcol + 64) * a.n_rows + row_t + 106] = accv[4][6];
                a.store[(qcol + 64) * a.n_rows + row_t + 107] = accv[4][7];
                a.store[(qcol + 64) * a.n_rows + row_t + 112] = accv[4][8];
                a.store[(qcol + 64) * a.n_rows + row_t + 113] = accv[4][9];
                a.store[(qcol + 64) * a.n_rows + row_t + 114] = accv[4][10];
                a.store[(qcol + 64) * a.n_rows + row_t + 115] = accv[4][11];
                a.store[(qcol + 64) * a.n_rows + row_t + 120] = accv[4][12];
                a.store[(qcol + 64) * a.n_rows + row_t + 121] = accv[4][13];
                a.store[(qcol + 64) * a.n_rows + row_t + 122] = accv[4][14];
                a.store[(qcol + 64) * a.n_rows + row_t + 123] = accv[4][15];
                a.store[(qcol + 96) * a.n_rows + row_t + 96] = accv[5][0];
                a.store[(qcol + 96) * a.n_rows + row_t + 97] = accv[5][1];
                a.store[(qcol + 96) * a.n_rows + row_t + 98] = accv[5][2];
                a.store[(qcol + 96) * a.n_rows + row_t + 99] = accv[5][3];
                a.store[(qcol + 96) * a.n_rows + row_t + 104] = accv[5][4];
                a.store[(qcol + 96) * a.n_rows + row_t + 105] = accv[5][5];
                a.store[(qcol + 96) * a.n_rows + row_t + 106] = accv[5][6];
                a.store[(qcol + 96) * a.n_rows + row_t + 107] = accv[5][7];
                a.store[(qcol + 96) * a.n_rows + row_t + 112] = accv[5][8];
                a.store[(qcol + 96) * a.n_rows + row_t + 113] = accv[5][9];
                a.store[(qcol + 96) * a.n_rows + row_t + 114] = accv[5][10];
                a.store[(qcol + 96) * a.n_rows + row_t + 115] = accv[5][11];
                a.store[(qcol + 96) * a.n_rows + row_t + 120] = accv[5][12];
                a.store[(qcol + 96) * a.n_rows + row_t + 121] = accv[5][13];
                a.store[(qcol + 96) * a.n_rows + row_t + 122] = accv[5][14];
                a.store[(qcol + 96) * a.n_rows + row_t + 123] = accv[5][15];
                a.store[(qcol + 128) * a.n_rows + row_t + 96] = accv[6][0];
                a.store[(qcol + 128) * a.n_rows + row_t + 97] = accv[6][1];
                a.store[(qcol + 128) * a.n_rows + row_t + 98] = accv[6][2];
                a.store[(qcol + 128) * a.n_rows + row_t + 99] = accv[6][3];
                a.store[(qcol + 128) * a.n_rows + row_t + 104] = accv[6][4];
                a.store[(qcol + 128) * a.n_rows + row_t + 105] = accv[6][5];
                a.store[(qcol + 128) * a.n_rows + row_t + 106] = accv[6][6];
                a.store[(qcol + 128) * a.n_rows + row_t + 107] = accv[6][7];
                a.store[(qcol + 128) * a.n_rows + row_t + 112] = accv[6][8];
                a.store[(qcol + 128) * a.n_rows + row_t + 113] = accv[6][9];
                a.store[(qcol + 128) * a.n_rows + row_t + 114] = accv[6][10];
                a.store[(qcol + 128) * a.n_rows + row_t + 115] = accv[6][11];
                a.store[(qcol + 128) * a.n_rows + row_t + 120] = accv[6][12];
                a.store[(qcol + 128) * a.n_rows + row_t + 121] = accv[6][13];
                a.store[(qcol + 128) * a.n_rows + row_t + 122] = accv[6][14];
                a.store[(qcol + 128) * a.n_rows + row_t + 123] = accv[6][15];
                a.store[(qcol + 160) * a.n_rows + row_t + 96] = accv[7][0];
                a.store[(qcol + 160) * a.n_rows + row_t + 97] = accv[7][1];
                a.store[(qcol + 160) * a.n_rows + row_t + 98] = accv[7][2];
                a.store[(qcol + 160) * a.n_rows + row_t + 99] = accv[7][3];
                a.store[(qcol + 160) * a.n_rows + row_t + 104] = accv[7][4];
                a.store[(qcol + 160) * a.n_rows + row_t + 105] = accv[7][5];
                a.store[(qcol + 160) * a.n_rows + row_t + 106] = accv[7][6];
                a.store[(qcol + 160) * a.n_rows + row_t + 107] = accv[7][7];
                a.store[(qcol + 160) * a.n_rows + row_t + 112] = accv[7][8];
                a.store[(qcol + 160) * a.n_rows + row_t + 113] = accv[7][9];
                a.store[(qcol + 160) * a.n_rows + row_t + 114] = accv[7][10];
                a.store[(qcol + 160) * a.n_rows + row_t + 115] = accv[7][11];
                a.store[(qcol + 160) * a.n_rows + row_t + 120] = accv[7][12];
                a.store[(qcol + 160) * a.n_rows + row_t + 121] = accv[7][13];
                a.store[(qcol + 160) * a.n_rows + row_t + 122] = accv[7][14];
                a.store[(qcol + 160) * a.n_rows + row_t + 123] = accv[7][15];
                ++ct;
            }
            if (++cks == KS2) cks = 0;
            cbuf = nbuf;
        }
        // keep the VGPR accumulators alive
        for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(accv[i]));
        __syncthreads();
    }
}

}  // namespace ccr

extern "C" int ccr_proto4w_time(const uint16_t *D, int64_t n_rows, int dim, const uint16_t *Q, int n_q, int ranges, int qgroups,
                                int reps, float *ms_out) {
    using namespace ccr;
    (void)qgroups;
    if (n_rows % TILE_DOCS || n_q % TQ || ranges % NUM_XCD) return -4;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.D = D; g.n_rows = n_rows; g.dim = dim; g.Q = Q; g.n_q = n_q;
    g.n_vt = n_rows / TILE_DOCS;
    g.tile_stride = 1;
    g.ranges = ranges;
    const size_t lds = RING3 * (size_t)SUBB;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm4w384_proto_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(gemm4w384_proto_kernel, dim3(256), dim3(P4_THREADS), lds, 0, g);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm4w384_proto_kernel, dim3(256), dim3(P4_THREADS), lds, 0, g);
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return -2;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// PROTO_MODE=3 only: one launch that writes the raw MFMA scores [n_q][n_rows] (fp32) of the whole problem
extern "C" int ccr_proto4w_scores(const uint16_t *D, int64_t n_rows, int dim, const uint16_t *Q, int n_q, int ranges, float *out) {
    using namespace ccr;
    if (PROTO_MODE != 3 || n_rows % TILE_DOCS || n_q % TQ || ranges % NUM_XCD) return -4;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.D = D; g.n_rows = n_rows; g.dim = dim; g.Q = Q; g.n_q = n_q;
    g.n_vt = n_rows / TILE_DOCS;
    g.tile_stride = 1;
    g.ranges = ranges;
    g.store = out;
    const size_t lds = RING3 * (size_t)SUBB;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm4w384_proto_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    hipLaunchKernelGGL(gemm4w384_proto_kernel, dim3(256), dim3(P4_THREADS), lds, 0, g);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
