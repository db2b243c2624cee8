// Orientation, blur and rBRIEF descriptor per selected keypoint, and the final
// keypoint record: IC_Angle (src/ORBextractor.cc:23-63), the 7x7 Gaussian blur
// the reference applies to the whole level before describing it (:491-493),
// computeOrbDescriptor (:67-103) and the tail of operator() (level scaling,
// level-major output order).
//
// One wavefront per keypoint.  The 43x43 source window around the keypoint is
// staged in LDS once; the 31x31 circular patch moments come from its centre,
// and the blur is evaluated only where the rotated sampling pattern reads it:
// horizontal sums for the window's rows, the vertical taps at the 512 sample
// points (all within radius 18.4 of the keypoint).  The blur is exact 8.8
// fixed point, so this equals blurring the whole level first.
#include <hip/hip_runtime.h>

#include <cfloat>

#include "orb_internal.h"

#define DESC_WAVES 1  // wavefronts per workgroup; with one, a keypoint's LDS is free again as soon as its wavefront ends
#define RAW_N 43
#define RAW_DW 12  // dwords fetched per window row: a row starting at any byte alignment fits
#ifndef RAW_P
#define RAW_P 48   // LDS pitch of the window rows
#endif
#define HGROUPS 10 // 4-column groups that hold the 37 blurred columns off+3 .. off+39, from group (off+3)/4 on
#define HROWS 6    // rows (or row pairs) the 64 lanes take per pass: lane = (row, group), lanes 60..63 idle
#ifndef HB_P
#define HB_P 40    // horizontal sums (16 bit) of a row pair per column of those groups: one dword each
#endif
#define HB_PAIRS 22  // row pairs (0,1) .. (42,43); row 43 does not exist, its sums meet weight 0 only
#define HB_FIRST 12  // pairs made by the first horizontal phase (source rows 0..23)
#define BL_N 37
#define HB_BYTES (HB_PAIRS * HB_P * 4)
#define RAW_OFF (HB_BYTES - 2 * HB_FIRST * RAW_P)
#define LDS_BYTES (RAW_OFF + (RAW_N + 1) * RAW_P)
static_assert(RAW_OFF >= HB_FIRST * HB_P * 4 && RAW_OFF % 16 == 0 && LDS_BYTES % 16 == 0,
              "the source window starts behind the first phase's sums, on a 16-byte boundary");

// the 256 test pairs of src/ORBextractor.cc:105-362 as (x0, x1, y0, y1), kept as floats for the rotation
__constant__ __attribute__((aligned(16))) float c_pattern[256 * 4] = {
#include "orb_pattern_xxyy.inc"
};

static __device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// cv::fastAtan2: every product / sum rounded on its own.
static __device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float s = (float)(180 / 3.14159265358979323846);
    const float p1 = __fmul_rn(0.9997878412794807f, s);
    const float p3 = __fmul_rn(-0.3258083974640975f, s);
    const float p5 = __fmul_rn(0.1555786518463281f, s);
    const float p7 = __fmul_rn(-0.04432655554792128f, s);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, (float)DBL_EPSILON));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, (float)DBL_EPSILON));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// Sine / cosine for the pattern rotation: reduce by multiples of pi/2 in double
// (two-constant Cody-Waite), fdlibm kernel polynomials in double, every
// operation rounded on its own, one rounding to float.  |x| < 2^20.
static __device__ __forceinline__ void sincos_portable(float xf, float* s, float* c) {
    const double x = (double)xf;
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00;
    const double pio2_1t = 6.07710050650619224932e-11;
    const double big = 6755399441055744.0;
    double fn = __dadd_rn(__dmul_rn(x, invpio2), big);
    fn = __dsub_rn(fn, big);
    const int n = (int)fn;
    double r = __dsub_rn(x, __dmul_rn(fn, pio2_1));
    r = __dsub_rn(r, __dmul_rn(fn, pio2_1t));
    const double z = __dmul_rn(r, r);
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double ps = S6;
    ps = __dadd_rn(__dmul_rn(ps, z), S5);
    ps = __dadd_rn(__dmul_rn(ps, z), S4);
    ps = __dadd_rn(__dmul_rn(ps, z), S3);
    ps = __dadd_rn(__dmul_rn(ps, z), S2);
    ps = __dadd_rn(__dmul_rn(ps, z), S1);
    const double sn = __dadd_rn(r, __dmul_rn(r, __dmul_rn(z, ps)));
    double pc = C6;
    pc = __dadd_rn(__dmul_rn(pc, z), C5);
    pc = __dadd_rn(__dmul_rn(pc, z), C4);
    pc = __dadd_rn(__dmul_rn(pc, z), C3);
    pc = __dadd_rn(__dmul_rn(pc, z), C2);
    pc = __dadd_rn(__dmul_rn(pc, z), C1);
    const double cs = __dadd_rn(__dsub_rn(1.0, __dmul_rn(0.5, z)), __dmul_rn(z, __dmul_rn(z, pc)));
    double so, co;
    switch (n & 3) {
        case 0: so = sn; co = cs; break;
        case 1: so = cs; co = -sn; break;
        case 2: so = -sn; co = -cs; break;
        default: so = -cs; co = sn; break;
    }
    *s = (float)so;
    *c = (float)co;
}

// lane / D for lane in 0..63 with a 24-bit multiply (the general 32-bit division by a constant costs a
// quarter-rate v_mul_hi_u32)
template <int D>
static constexpr bool lane_div_ok() {
    for (int l = 0; l < 64; l++)
        if (((l * (512 / D + 1)) >> 9) != l / D) return false;
    return true;
}
template <int D>
static __device__ __forceinline__ int lane_div(int lane) {
    static_assert(lane_div_ok<D>(), "multiplier does not reproduce lane / D");
    return (int)(__umul24((unsigned)lane, 512 / D + 1) >> 9);
}

static __device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef float float2v __attribute__((ext_vector_type(2)));  // element-wise products and sums: v_pk_mul_f32 / v_pk_add_f32
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
// sum of the two 16-bit products of a and b, plus c (v_dot2_u32_u16)
static __device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b), c, false);
}
// sum of the four byte products of a and b, plus c (v_dot4_u32_u8)
static __device__ __forceinline__ uint32_t dot4_u8(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot4(a, b, c, false);
}

// Sum over the 64 lanes, returned to all of them: prefix sums inside each row of 16 lanes and two row
// broadcasts through the DPP path (lanes a step does not feed add 0), then a read of lane 63.
template <int CTRL, int ROWS>
static __device__ __forceinline__ int dpp_add_step(int v) {
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROWS, 0xf, false);
}
static __device__ __forceinline__ int wave_sum(int v) {
    v = dpp_add_step<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add_step<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add_step<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add_step<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of every row holds the row's sum
    v = dpp_add_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// LEGACY: the Gaussian taps of OpenCV before 3.x (ORB_Q_GAUSS_LEGACY), which sum to 257 instead of 256.
template <bool LEGACY>
__global__ __launch_bounds__(64 * DESC_WAVES) void k_describe(const OrbGeom* __restrict__ g,
                                                             const uint8_t* __restrict__ img0, int stride0,
                                                             int64_t frameStride0, const uint8_t* __restrict__ pyr,
                                                             int frame0, int nframes,
                                                             const uint32_t* __restrict__ selXY,
                                                             const int* __restrict__ selScore,
                                                             const int* __restrict__ levelCount,
                                                             orb_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                             int capPerFrame, int* __restrict__ nOut,
                                                             int* __restrict__ error, int* __restrict__ status) {
    // One block per wavefront: the horizontal sums of the 22 row pairs (HB_BYTES), and the source window (44 rows: + the
    // partner row of row 42) placed so that its rows 0..23 are the tail of that region and rows 24..43 lie behind it --
    // the sums of pairs 12..21, made from rows 24..43, overwrite rows the pairs 0..11 have consumed.
    __shared__ __attribute__((aligned(16))) uint8_t s_lds[DESC_WAVES][LDS_BYTES];
    // the wavefront index is the same in all 64 lanes: telling the compiler so keeps everything derived from
    // it (frame, level, keypoint, window addresses, loop bounds) in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int outCapFrame = g->outCapFrame, nlevels = g->nlevels;
    // one keypoint slot per workgroup, a frame's slots on one XCD (its windows overlap: they meet in that L2)
    static_assert(DESC_WAVES == 1, "the slot numbering assumes one wavefront per workgroup");
    int frameLocal;
    unsigned slotU;
    if (!orb_xcd_map(blockIdx.x, (unsigned)outCapFrame, g->outCapFrameMagic, nframes, &frameLocal, &slotU)) return;
    const int frame = frame0 + frameLocal;
    const int slot = (int)slotU;
    int level = 0;
    for (int l = 1; l < nlevels; l++)
        if (slot >= g->lv[l].outBase) level = l;
    const OrbLevel& L = g->lv[level];
    const int i = slot - L.outBase;
    const int* lc = levelCount + frame * nlevels;
    int before = 0, total = 0;
    for (int l = 0; l < nlevels; l++) {
        const int cnt = lc[l];
        if (l < level) before += cnt;
        total += cnt;
    }
    if (slot == 0 && lane == 0) {
        nOut[frame] = total;
        if (total > capPerFrame) atomicOr(error, 64);  // caller's buffers too small: reported, rows beyond cap dropped
        // Callers that read their results without a copy of the error word get it per frame: everything the earlier
        // stages flagged (they are complete) and this frame's own overflow.
        if (status) status[frame] = __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | (total > capPerFrame ? 64 : 0);
    }
    if (i >= lc[level]) return;
    const int outIdx = before + i;
    if (outIdx >= capPerFrame) return;

    const uint32_t xy = selXY[(int64_t)frame * outCapFrame + slot];
    const int score = selScore[(int64_t)frame * outCapFrame + slot];
    const int px = (int)(xy & 0xffff) + ORB_BORDER, py = (int)(xy >> 16) + ORB_BORDER;  // :885-886

    const uint8_t* img;
    int pitch;
    if (level == 0) {
        img = img0 + (int64_t)frame * frameStride0;
        pitch = stride0;
    } else {
        img = pyr + L.off + (int64_t)frame * L.frameStride;
        pitch = L.pitch;
    }
    uint8_t* raw = s_lds[wave] + RAW_OFF;
    uint32_t* hb = (uint32_t*)s_lds[wave];

    // 43x43 source window.  On 4-byte-aligned rows it is fetched as whole dwords (12 per row, the window then starts
    // `off` bytes into the LDS row): interior keypoints as they are; keypoints whose window leaves the level by its
    // one or two rows or columns (a keypoint is at least 19 pixels from the edge, the window reaches 21) with the row
    // numbers reflected like the blur's BORDER_REFLECT_101, dwords beyond the row fetched from its first or last
    // dword instead, and the missing columns copied from their mirror images inside the window afterwards.  Rows
    // off 4-byte boundaries take bytes.
    const int wx0 = px - 21, wy0 = py - 21;
    const int xa = wx0 & ~3;
    const bool dwordRows = (pitch & 3) == 0 && (((unsigned long long)img) & 3ull) == 0;
    // (on the library's own levels the last dwords of a row may lie beyond the level's width, inside the row pitch:
    // bytes that are no pixel of the window; the caller's level 0 is not read beyond its width)
    const int rowLimit = level == 0 ? L.w : pitch;
    const bool fastLoad = dwordRows && wx0 >= 0 && wy0 >= 0 && py + 21 < L.h && px + 21 < L.w && xa + 4 * RAW_DW <= rowLimit;
    const bool borderLoad = dwordRows && !fastLoad && (rowLimit & 3) == 0 && rowLimit >= 4 * RAW_DW && L.h >= 4;
    const int off = (fastLoad || borderLoad) ? (wx0 - xa) : 0;
    // the IC_Angle byte weights of this lane's five patch rows: requested now, used after the window is staged
    const int icRow = lane_div<9>(lane), icGroup = lane - icRow * 9;
    uint2 icm[5];
    if (icRow < 7) {
        const uint2* mk = (const uint2*)&g->icMask[off][icRow][icGroup][0];
#pragma unroll
        for (int it = 0; it < 5; it++)
            if (icRow + 7 * it < 31) icm[it] = mk[it * 7 * 9];
    }
    if (fastLoad) {
        // lane = (row within a group of five, dword); nine groups cover the 43 rows, all loads in flight at once.
        // The window's origin is the same for all lanes (scalar 64-bit address), a lane adds a 32-bit offset.
        const int rr = lane_div<RAW_DW>(lane), d = lane - rr * RAW_DW;
        if (rr < 5) {
            // (the stride of five rows goes through a lane read so that it stays a plain addend: folded into
            // the address it becomes one quarter-rate 64-bit multiply-add per row)
            const int64_t step = (int64_t)__builtin_amdgcn_readfirstlane(5 * pitch);
            const uint8_t* base = img + (int64_t)wy0 * pitch + xa +
                                  (__umul24((unsigned)rr, (unsigned)pitch) + 4u * (unsigned)d);
            uint32_t* dstw = (uint32_t*)raw + rr * (RAW_P / 4) + d;
            uint32_t v[9];
#pragma unroll
            for (int it = 0; it < 9; it++)
                if (rr + 5 * it < RAW_N) v[it] = *(const uint32_t*)(base + it * step);
#pragma unroll
            for (int it = 0; it < 9; it++)
                if (rr + 5 * it < RAW_N) dstw[it * 5 * (RAW_P / 4)] = v[it];
        }
    } else if (borderLoad) {
        const int rr = lane_div<RAW_DW>(lane), d = lane - rr * RAW_DW;
        if (rr < 5) {
            const int colx = min(max(xa + 4 * d, 0), rowLimit - 4);
            uint32_t* dstw = (uint32_t*)raw + rr * (RAW_P / 4) + d;
            uint32_t v[9];
#pragma unroll
            for (int it = 0; it < 9; it++) {
                const int y = wy0 + rr + 5 * it;
                const int sy = y < 0 ? -y : (y >= L.h ? 2 * L.h - 2 - y : y);
                if (rr + 5 * it < RAW_N) v[it] = *(const uint32_t*)(img + (int64_t)sy * pitch + colx);
            }
#pragma unroll
            for (int it = 0; it < 9; it++)
                if (rr + 5 * it < RAW_N) dstw[it * 5 * (RAW_P / 4)] = v[it];
        }
        const int nLeft = wx0 < 0 ? -wx0 : 0, nRight = wx0 + RAW_N > L.w ? wx0 + RAW_N - L.w : 0;  // at most 2 each
        if (nLeft | nRight) {
            wave_lds_sync();
            if (lane < RAW_N) {
                uint8_t* row = raw + lane * RAW_P + off;
                for (int c = 0; c < nLeft; c++) row[c] = row[-2 * wx0 - c];                    // x = wx0 + c -> -x
                for (int c = RAW_N - nRight; c < RAW_N; c++) row[c] = row[2 * (L.w - 1 - wx0) - c];  // x -> 2 (w - 1) - x
            }
        }
    } else {
        for (int k = lane; k < RAW_N * RAW_N; k += 64) {
            const int r = k / RAW_N, c = k - r * RAW_N;
            const int sx = reflect101(wx0 + c, L.w), sy = reflect101(wy0 + r, L.h);
            raw[r * RAW_P + c] = img[(int64_t)sy * pitch + sx];
        }
    }
    wave_lds_sync();

    // Intensity centroid over the circular patch (rows v, half-width umax[|v|]): rows v = -15..15, aligned column
    // groups 1..9 (they cover u = -15..15 for every `off`); lane = (row within a pass of seven, group), five passes
    // cover the 31 rows.  Which bytes of a dword lie inside the patch row, and their weights u + 32 (positive for
    // every dword a lane can hold; taken off again through the plain sum m00: sum u * I = sum (u + 32) * I -
    // 32 * sum I), come from the table the host built for the context's umax (icm, fetched above).
    int m10 = 0, m01 = 0;
    {
        const uint32_t* raw32 = (const uint32_t*)raw;
        int m00 = 0;
        if (icRow < 7) {
#pragma unroll
            for (int it = 0; it < 5; it++) {
                const int r = icRow + 7 * it;
                if (r < 31) {
                    const uint32_t w = raw32[(6 + r) * (RAW_P / 4) + 1 + icGroup];
                    const int rowSum = (int)dot4_u8(w, icm[it].x, 0u);
                    m10 = (int)dot4_u8(w, icm[it].y, (uint32_t)m10);
                    m00 += rowSum;
                    m01 += __mul24(r - 15, rowSum);
                }
            }
        }
        m10 -= 32 * m00;
    }
    m10 = wave_sum(m10);
    m01 = wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);

    // Blur of the 37x37 window the pattern can reach, evaluated where the pattern samples it.  The horizontal pass
    // works on the aligned columns of the LDS rows, four at a time from whole dwords, and leaves 8.8 sums (16 bit) for
    // all 44 rows; the vertical taps are then applied at the 512 sample points only (two per lane and round), seven
    // rows of sums against the taps with v_dot2_u32_u16 and one rounding -- the value the blurred level has there.
    // Aligned columns outside [off + 3, off + 39] hold sums of bytes that are not part of the window; nothing reads them.
    constexpr uint32_t k0 = 18, k1 = 34, k2 = LEGACY ? 49 : 48, k3 = LEGACY ? 55 : 56;
    // taps k0 k1 k2 k3 k2 k1 k0 against the bytes of the dwords A (columns 4q-4..4q-1), B (4q..4q+3), C (4q+4..4q+7)
    // for the outputs at columns 4q + 0..3
    const uint32_t wA0 = (k0 << 8) | (k1 << 16) | (k2 << 24), wB0 = k3 | (k2 << 8) | (k1 << 16) | (k0 << 24);
    const uint32_t wA1 = (k0 << 16) | (k1 << 24), wB1 = k2 | (k3 << 8) | (k2 << 16) | (k1 << 24), wC1 = k0;
    const uint32_t wA2 = k0 << 24, wB2 = k1 | (k2 << 8) | (k3 << 16) | (k2 << 24), wC2 = k1 | (k0 << 8);
    const uint32_t wB3 = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24), wC3 = k2 | (k1 << 8) | (k0 << 16);
    const uint32_t* raw32 = (const uint32_t*)raw;
    uint32_t* hb32 = hb;
    // lane = (row pair within a pass of six, group of four columns); lanes 60..63 idle.  The horizontal
    // sums of rows 2p and 2p + 1 are stored side by side, one dword per column: seven consecutive rows are then
    // four such dwords whatever the parity of the first one.
    const int rr = lane_div<HGROUPS>(lane), qi = lane - rr * HGROUPS;
    const int q = ((off + 3) >> 2) + qi;  // the group's place in the LDS rows of the source window
    // LDS operations of the one wavefront execute in program order; the syncs only keep the compiler from moving
    // them across.
    const int back = q > 0 ? -1 : 0;
    auto hpass = [&](int pr) {
        const uint32_t* p = raw32 + 2 * pr * (RAW_P / 4) + q;
        uint32_t o[2][4];
#pragma unroll
        for (int h = 0; h < 2; h++, p += RAW_P / 4) {
            const uint32_t A = p[back], B = p[0], C = p[1];
            // output m of the group sums bytes m-3 .. m+3 around column 4q + m: the seven taps laid over
            // the three aligned dwords that hold them (no byte alignment of the data)
            o[h][0] = dot4_u8(A, wA0, dot4_u8(B, wB0, 0u));
            o[h][1] = dot4_u8(A, wA1, dot4_u8(B, wB1, dot4_u8(C, wC1, 0u)));
            o[h][2] = dot4_u8(A, wA2, dot4_u8(B, wB2, dot4_u8(C, wC2, 0u)));
            o[h][3] = dot4_u8(B, wB3, dot4_u8(C, wC3, 0u));
        }
        uint4 e;  // (sums stay below 2^16)
        e.x = (o[1][0] << 16) + o[0][0];
        e.y = (o[1][1] << 16) + o[0][1];
        e.z = (o[1][2] << 16) + o[0][2];
        e.w = (o[1][3] << 16) + o[0][3];
        *(uint4*)(hb32 + pr * HB_P + 4 * qi) = e;
    };
    static_assert(HROWS == 6 && HB_PAIRS == 22 && HB_FIRST == 12 && BL_N == 37, "the phases below are laid out for these");
    if (rr < HROWS) {
        hpass(rr);           // pairs 0..5
        hpass(rr + 6);       // pairs 6..11
    }
    wave_lds_sync();         // source rows 0..23 are consumed: the sums of pairs 12..21 may take their place
    if (rr < HROWS) {
        hpass(rr + 12);      // pairs 12..17
        if (rr + 18 < HB_PAIRS) hpass(rr + 18);  // pairs 18..21
    }
    wave_lds_sync();

    // rBRIEF: lane l of round r compares pair 64r + l; the ballot is descriptor bytes 8r..8r+7.
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float a, b;
    sincos_portable(__fmul_rn(angle, factorPI), &b, &a);
    // Rounding to the nearest integer, ties to even (cvRound), by adding 1.5 * 2^23: the sum's bits are those of the
    // constant plus the rounded value, for any |value| < 2^22.  Window row and column of a sample point are those bits
    // less (the constant's - 18).
    const float magic = 12582912.f;
    const uint32_t magicBits = 0x4B400000u;
    // tap pairs against (row 2p, row 2p + 1) for an even first row r = 2p, and for an odd one r = 2p + 1
    const uint32_t wE0 = k0 | (k1 << 16), wE1 = k2 | (k3 << 16), wE2 = k2 | (k1 << 16), wE3 = k0;
    const uint32_t wO0 = k0 << 16, wO1 = k1 | (k2 << 16), wO2 = k3 | (k2 << 16), wO3 = k1 | (k0 << 16);
    // Byte address of the sums of (row pair p, window column c): with yb and xb the bits of the rounded floats,
    // row r = yb - (magicBits - 18) is the first of the seven rows and p = r >> 1; the constant is even, so r has
    // yb's parity, and the low 24 bits of yb >> 1 are 0xA00000 - 9 + p: one shift, one 24-bit multiply-add and one
    // shift-add, the constants folded into `base`.
    const uint32_t base = (uint32_t)(uintptr_t)(lds_u8*)hb + 4u * (uint32_t)((off + 3) & 3) - 4u * (magicBits - 18u) -
                          (0xA00000u - 9u) * (4u * HB_P);
    auto blurred_at = [&](uint32_t yBits, uint32_t xBits) -> uint32_t {
        const bool odd = (yBits & 1u) != 0u;
        const uint32_t addr = __umul24(yBits >> 1, 4u * HB_P) + ((xBits << 2) + base);
        const lds_u32* p = (const lds_u32*)(uintptr_t)addr;
        uint32_t v = dot2_u16(p[0], odd ? wO0 : wE0, 32768u);
        v = dot2_u16(p[HB_P], odd ? wO1 : wE1, v);
        v = dot2_u16(p[2 * HB_P], odd ? wO2 : wE2, v);
        v = dot2_u16(p[3 * HB_P], odd ? wO3 : wE3, v);
        return LEGACY ? min(v >> 16, 255u) : v >> 16;  // (taps that sum to 257: the rounded value can pass 255)
    };
    unsigned long long words[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int pair = 64 * r + lane;
        const float4 pt = ((const float4*)c_pattern)[pair];
        // both test points at once: (x0, x1) and (y0, y1) rotate to y = x*b + y*a, x = x*a - y*b, each product
        // and sum rounded on its own (v_pk_mul_f32 / v_pk_add_f32 are element-wise)
        const float2v xs = {pt.x, pt.y}, ys = {pt.z, pt.w};
        const float2v fy = (xs * b + ys * a) + magic;
        const float2v fx = (xs * a - ys * b) + magic;
        const uint32_t t0 = blurred_at(__float_as_uint(fy.x), __float_as_uint(fx.x));
        const uint32_t t1 = blurred_at(__float_as_uint(fy.y), __float_as_uint(fx.y));
        words[r] = __ballot(t0 < t1);
    }
    if (lane == 0) {
        const int64_t o = (int64_t)frame * capPerFrame + outIdx;
        unsigned long long* d = (unsigned long long*)(desc + o * 32);
        d[0] = words[0];
        d[1] = words[1];
        d[2] = words[2];
        d[3] = words[3];
        orb_keypoint kp;
        kp.x = __fmul_rn((float)px, L.scale);   // level 0: scale is 1
        kp.y = __fmul_rn((float)py, L.scale);
        kp.size = L.patchSize;
        kp.angle = angle;
        kp.response = (float)score;
        kp.octave = level;
        kp.class_id = -1;
        kps[o] = kp;
    }
}

int orb_launch_describe(const OrbBatch& b) {
    const OrbGeom& g = *b.h_geom;
    const int64_t blocks = orb_xcd_grid(b.nframes, g.outCapFrame);
    if (g.quirks & ORB_Q_GAUSS_LEGACY)
        hipLaunchKernelGGL(k_describe<true>, dim3((unsigned)blocks), dim3(64 * DESC_WAVES), 0, (hipStream_t)b.stream, b.d_geom,
                           b.d_img0, b.stride0, b.frameStride0, b.d_pyr, b.frame0, b.nframes, b.d_selXY, b.d_selScore,
                           b.d_levelCount, b.d_kps, b.d_desc, b.capPerFrame, b.d_nOut, b.d_error, b.d_status);
    else
        hipLaunchKernelGGL(k_describe<false>, dim3((unsigned)blocks), dim3(64 * DESC_WAVES), 0, (hipStream_t)b.stream, b.d_geom,
                           b.d_img0, b.stride0, b.frameStride0, b.d_pyr, b.frame0, b.nframes, b.d_selXY, b.d_selScore,
                           b.d_levelCount, b.d_kps, b.d_desc, b.capPerFrame, b.d_nOut, b.d_error, b.d_status);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        orb_set_error("describe launch failed: %s", hipGetErrorString(e));
        return ORB_ERR_HIP;
    }
    return ORB_OK;
}
