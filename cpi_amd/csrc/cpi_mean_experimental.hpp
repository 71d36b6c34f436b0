// cpi_mean_experimental.hpp -- measurement-only mean kernels (LDS-DMA ring, block-resident linear fetch): DESIGN.md 3.1, never the default path.
// Part of the translation unit cpi_mean.hip (CPI_EXPERIMENTS builds only) (included there after cpi_math.hpp / cpi_device_util.hpp; not a stand-alone header).
#pragma once

namespace {

// ============================================================================================
// mean kernel, large batches: knots streamed into LDS by the DMA path (global_load_lds_dwordx4)
// ============================================================================================
// One lane per window, 64 consecutive windows per wavefront (dense layout).  A STAGE is KC knots of every window of
// the wavefront: 64 x KC x 56 B.  Stages land in a ring of S LDS slots through LDS-DMA loads, i.e. without passing
// through (and without costing) vector registers: the cpi_mean_kernel stage of 14 doubles + 14 pointers + 42 address
// words per lane is gone, the prefetch distance is S - 1 whole stages, and a window contributes KC x 56 contiguous
// bytes per request instead of 112 (DRAM page locality: a pure-read kernel with this access pattern streams 2.86 GB in
// 0.53 ms with 112-byte pieces, 0.48 ms with 448-byte pieces, 0.44 ms linearly -- DESIGN.md 3.1).
//
// LDS-DMA writes "wave-uniform base (M0) + 16 x lane", so the LDS image of one DMA instruction is fixed: lane l's 16
// bytes at 16 l.  Lane l of instruction j fetches piece (l mod PPW) of window j*WPI + l / PPW of the block: the PPW
// lanes of a window read PPW x 16 contiguous bytes (coalesced), WPI = 64 / PPW windows per instruction, 64 mod PPW
// idle lanes re-fetch a valid address.  ALIGNED: every piece is fetched from a 16-byte aligned address -- a window
// starts on an 8-byte boundary only (56-byte knots), so a stage is fetched as the aligned superset of PPW = KC*3.5 + 1
// pieces and the reader skips its window's leading 0 / 8 bytes.  The odd piece count also spreads the readers' rows over
// the LDS banks (row pitch 240 B -> 2-way conflicts on the 8-byte reads; 224 B would be 4-way, 256 B 32-way).
// The global address of instruction j is "scalar base (SALU) + constant 32-bit lane offset": no vector address
// arithmetic at all.  Ordering: the wave's own counted s_waitcnt vmcnt is what orders its ds_reads behind its LDS-DMA
// (MI355X_MICROARCH.md item 7; single-wave workgroup, no barrier needed); a slot is re-armed only after an
// lgkmcnt(0) has retired the reads of its previous contents.
// MEASURED (MI355X, profiles/r02_mean_lds_dma.md; 1 M x 50, cpi_mean_kernel = 0.669 ms): KC,S = 4,2 unaligned 0.675 ms,
// 4,2 aligned 0.685, 2,3 0.72, 4,3 0.71, 8,1 0.75, 8,2 (2 wavefronts per CU) 1.03; 100 k x 50: 0.119 vs 0.073 ms (5
// wavefronts per CU by LDS = two rounds of wavefronts instead of one).  The staging registers are gone (110 VGPRs instead
// of 212) and 16 KB per wavefront are in flight, yet nothing is gained: the strided multi-stream pattern itself delivers
// ~4.8 TB/s of DMA traffic.  NOT the default: reachable through CPI_AMD_MEAN_DMA=KC,S,A for A/B measurements only.
template <int KC, bool ALIGNED>
struct DmaGeom {
    static constexpr int PPW = (KC * 56) / 16 + (ALIGNED ? 1 : 0);   // 16-byte pieces per window and stage
    static constexpr int WPI = 64 / PPW;                             // windows per DMA instruction
    static constexpr int NI = (64 + WPI - 1) / WPI;                  // DMA instructions per stage
    static constexpr int SLOT = NI * 1024;                           // bytes per ring slot
    static_assert((KC * 56) % 16 == 0, "a stage is a whole number of 16-byte pieces (KC even)");
    static_assert(!ALIGNED || (WPI % 2 == 0), "alignment phase of an instruction's first window must not depend on j");
};
// (glds16 / wait_vmcnt: cpi_device_util.hpp)
template <int MODEL, bool AVG, int KC, int S, bool ALIGNED>
__global__ __launch_bounds__(64, 1) void cpi_mean_dma_kernel(PreArgs A) {
    typedef DmaGeom<KC, ALIGNED> G;
    static_assert((S - 1) * G::NI <= 63, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(1024))) char ring[S * G::SLOT];
    const int lane = threadIdx.x;
    const long long w = (long long)blockIdx.x * 64 + lane;          // every window of a block exists (host guarantees)
    const int n = A.count ? min(max(A.count[w], 0), A.N) : A.N;
    const int nmax = __builtin_amdgcn_readfirstlane(wave_max(n));
    const long long wstride = (long long)(A.N + 1) * 56;            // bytes per window
    const char *blk = reinterpret_cast<const char *>(A.knots) + (long long)blockIdx.x * 64 * wstride;   // wave-uniform

    // ---- DMA role of this lane (constant over instructions and stages)
    const int dw = lane / G::PPW, dp = lane - dw * G::PPW;          // window within the instruction, piece within the window
    const bool idle = dw >= G::WPI;
    // alignment phase: byte address of knot 1 of window (instr j, dw) = blk + (j WPI + dw) wstride + 56 + stage offset;
    // KC and WPI even -> its bit 3 depends on dw alone
    unsigned voff, rshift = 0;
    {
        const unsigned long long a0 = (unsigned long long)(blk + 56);
        const unsigned ph = ALIGNED ? (unsigned)((a0 + (unsigned long long)(idle ? 0 : dw) * (unsigned long long)wstride) & 8ull) : 0u;
        // scalar base is biased by -16 so that the lane offset stays non-negative
        voff = (unsigned)((idle ? 0 : dw) * wstride) + 16u * (unsigned)(idle ? 0 : dp) + 16u - ph;
        if (ALIGNED) {
            const int rw = lane % G::WPI;                            // this lane's OWN window sits at row rw of instruction lane / WPI
            rshift = (unsigned)((a0 + (unsigned long long)rw * (unsigned long long)wstride) & 8ull);
        }
    }
    const unsigned ring_base = (unsigned)(size_t)((__attribute__((address_space(3))) char *)ring);
    const int rd_off = (lane / G::WPI) * 1024 + (lane % G::WPI) * (G::PPW * 16) + (int)rshift;   // reader: own window's row

    const int nst = (nmax + KC - 1) / KC;
    const int dbg = A.dbg;   // development: 1 = no arithmetic, 2 = no fetch
    auto issue = [&](int st) {
        if (dbg & 2) return;
        const char *sb = blk + 56 - 16 + (long long)st * (KC * 56);
        const unsigned dst = ring_base + (unsigned)(st % S) * G::SLOT;
#pragma unroll
        for (int j = 0; j < G::NI; ++j) glds16(voff, sb + (long long)j * G::WPI * wstride, dst + j * 1024);
    };

    // prologue: S - 1 stages in flight, then this lane's first knot and linearisation point through ordinary loads
#pragma unroll
    for (int p = 0; p < S - 1; ++p) if (p < nst) issue(p);
    V3 bw = ldv3(A.lin + w * 6), ba = ldv3(A.lin + w * 6 + 3);
    V3 gk = mk(0, 0, 0);
    if (MODEL == 2) gk = mul(quat_2_Rot(ldq4(A.qk + w * 4)), mk(A.grav[0], A.grav[1], A.grav[2]));
    double pk[7];
    {
        const double *kb = A.knots + w * (long long)(A.N + 1) * 7;
#pragma unroll
        for (int i = 0; i < 7; i++) pk[i] = kb[i];
    }
    MeanState<false> st_;
    mean_init(st_);
    // The ordinary loads above must be complete BEFORE the loop: hipcc would otherwise wait for them at their first
    // use inside it -- an s_waitcnt vmcnt(0) executed in every iteration, which also drains the prefetched stages
    // (its counter model does not include the LDS-DMA instructions).
#pragma unroll
    for (int i = 0; i < 7; i++) asm volatile("" : "+v"(pk[i]));
    asm volatile("" : "+v"(bw.x), "+v"(bw.y), "+v"(bw.z), "+v"(ba.x), "+v"(ba.y), "+v"(ba.z));
    asm volatile("" : "+v"(gk.x), "+v"(gk.y), "+v"(gk.z));

    for (int st = 0; st < nst; ++st) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // reads of the slot about to be re-armed have retired
        const int ahead = min(S - 1, nst - 1 - st);                  // stages younger than st that are (or get) in flight
        if (S > 1 && st + S - 1 < nst) issue(st + S - 1);
        if (S == 1) issue(st);
        // stage st has landed once at most `ahead` stages' worth of younger DMA instructions are outstanding
        if (S == 1 || ahead == 0) wait_vmcnt<0>();
        else if (ahead == 1) wait_vmcnt<(S > 1 ? 1 : 0) * G::NI>();
        else if (ahead == 2) wait_vmcnt<(S > 2 ? 2 : 0) * G::NI>();
        else wait_vmcnt<(S > 3 ? 3 : 0) * G::NI>();
        const char *slot = ring + (st % S) * G::SLOT + rd_off;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int s = st * KC + c;
            if (s >= nmax || (dbg & 1)) break;                       // wave-uniform
            const double *nk = reinterpret_cast<const double *>(slot + 56 * c);
            double q[7];
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = nk[i];
            mean_step<MODEL, false, AVG>(st_, pk[0], q[0], mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                         mk(q[1], q[2], q[3]), mk(q[4], q[5], q[6]), bw, ba, gk, s < n);
#pragma unroll
            for (int i = 0; i < 7; i++) pk[i] = q[i];
        }
    }

    if (A.out.DT) A.out.DT[w] = st_.DT;
    if (A.out.alpha) stv3(A.out.alpha + w * 3, st_.alpha);
    if (A.out.beta) stv3(A.out.beta + w * 3, st_.beta);
    if (A.out.q) {
        const Q4 q = rot_2_quat(st_.R);
        double *p = A.out.q + w * 4;
        p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w;
    }
}

// ============================================================================================
// mean kernel, block-resident: a wavefront owns 64 / L CONSECUTIVE WHOLE windows (dense layout)
// ============================================================================================
// The wavefront's windows are one contiguous byte range of the knot array (64/L x (N+1) x 56 B -- 22.8 KB for
// L = 8, N = 50).  It is fetched LINEARLY by LDS-DMA, 1 KiB per instruction, every 128-byte line exactly once, and
// lands in LDS as the exact memory image; nothing passes through registers and there is no per-chunk dependency on
// the memory system (cpi_mean_kernel walks 9 serial chunks at 10 k windows).  Lane l of a window then integrates its
// contiguous run of ceil(n / L) intervals straight out of LDS and the L segments are composed by the order-preserving
// tree of cpi_mean_kernel (DPP row shifts: no LDS crossbar).  LDS per wavefront is dynamic (= the block's bytes):
// 7 wavefronts share a CU at N = 50 and overlap each other's fetch and arithmetic.
// MEASURED (MI355X, 1 M x 50, profiles/r02_mean_lds_dma.md): the fetch alone runs at 6.1 TB/s (0.48 ms), the arithmetic
// alone takes 0.53 ms (7 intervals + 3 tree levels per lane: ~45 % more FP64 instructions than one lane per window),
// together 0.675 ms -- the same as cpi_mean_kernel (0.67 ms), whose strided pattern is slower to fetch but whose
// arithmetic is minimal.  A persistent variant (buffer re-armed under the tree) was slower (0.72 ms).  NOT the default:
// reachable through CPI_AMD_MEAN_BLK=L for A/B measurements only.
// zero-filling row shift (lanes whose source lies outside the 16-lane row read 0): one v_mov_b32_dpp per half, no copy
template <int CTRL>
__device__ __forceinline__ double dpp_mov64z(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true),
                            __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ double row16_down0(double v, int d) {    // lane j <- lane j + d, 0 past the end of the row
    switch (d) {
        case 1: return dpp_mov64z<0x101>(v);
        case 2: return dpp_mov64z<0x102>(v);
        case 4: return dpp_mov64z<0x104>(v);
        default: return dpp_mov64z<0x108>(v);
    }
}
template <int MODEL, bool AVG, int L>
__global__ __launch_bounds__(64, 1) void cpi_mean_blk_kernel(PreArgs A) {
    static_assert((L & (L - 1)) == 0 && L >= 2 && L <= 32, "L lanes per window, power of two");
    constexpr int WPB = 64 / L;
    extern __shared__ __attribute__((aligned(1024))) char blkmem[];
    const int lane = threadIdx.x;
    const int grp = lane / L, l = lane % L;
    const long long w0 = (long long)blockIdx.x * WPB;
    const bool valid = (w0 + grp) < A.W;
    const long long w = valid ? w0 + grp : A.W - 1;
    const int nwin = (int)min((long long)WPB, A.W - w0);            // windows of this block (wave-uniform)
    const int wstride = (A.N + 1) * 56;                             // bytes per window (host guarantees WPB * wstride <= 64 KB)
    const int total = nwin * wstride;
    const char *blk = reinterpret_cast<const char *>(A.knots) + w0 * (long long)wstride;
    const unsigned ring_base = (unsigned)(size_t)((__attribute__((address_space(3))) char *)blkmem);
    const int dbg = A.dbg;   // development: 1 = no arithmetic, 2 = no fetch (where the time of this design goes)

    // ---- linear fetch of the block: instruction j moves bytes [1024 j, 1024 j + 1024)
    if (!(dbg & 2)) {
        const int nfull = total >> 10;
        const unsigned v16 = 16u * (unsigned)lane;
        for (int j = 0; j < nfull; ++j) glds16(v16, blk + ((long long)j << 10), ring_base + ((unsigned)j << 10));
        // the last, partial instruction runs with the lanes past the end of the block masked off (no load, no LDS write)
        const unsigned off = (unsigned)(nfull << 10) + v16;
        if ((int)(off + 16u) <= total) glds16(off, blk, ring_base + ((unsigned)nfull << 10));
    }
    const int n = valid ? (A.count ? min(max(A.count[w], 0), A.N) : A.N) : 0;
    const int per = (n + L - 1) / L;
    const int s0 = min(n, l * per), s1 = min(n, s0 + per);
    const int len = s1 - s0;
    const int maxlen = (dbg & 1) ? 0 : __builtin_amdgcn_readfirstlane(wave_max(len));
    V3 bw = ldv3(A.lin + w * 6), ba = ldv3(A.lin + w * 6 + 3);
    V3 gk = mk(0, 0, 0);
    if (MODEL == 2) gk = mul(quat_2_Rot(ldq4(A.qk + w * 4)), mk(A.grav[0], A.grav[1], A.grav[2]));
    // a block of 8-byte-odd length ends in the middle of a 16-byte piece: its last double comes through a register
    double lastd = 0.0;
    const bool patch = (total & 15) != 0;
    if (patch) lastd = *reinterpret_cast<const double *>(blk + total - 8);
    asm volatile("" : "+v"(bw.x), "+v"(bw.y), "+v"(bw.z), "+v"(ba.x), "+v"(ba.y), "+v"(ba.z), "+v"(lastd));
    asm volatile("" : "+v"(gk.x), "+v"(gk.y), "+v"(gk.z));
    wait_vmcnt<0>();                                                // the wave's own counted wait orders its ds_reads behind its LDS-DMA
    if (patch && lane == 0) *reinterpret_cast<double *>(blkmem + total - 8) = lastd;
    wave_lds_fence();

    constexpr bool GSEG = (MODEL == 2);
    MeanState<false> st;
    mean_init(st);
    GravAcc ga;
    if (GSEG) grav_init(ga);
    const double *kp = reinterpret_cast<const double *>(blkmem + (valid ? grp : 0) * wstride) + (long long)s0 * 7;
    double pk[7];
#pragma unroll
    for (int i = 0; i < 7; i++) pk[i] = kp[i];
    for (int s = 0; s < maxlen; ++s) {
        const double *nk = kp + 7 * min(s + 1, len);                 // knot s0 + len is the segment's last: always inside the block
        double q[7];
#pragma unroll
        for (int i = 0; i < 7; i++) q[i] = nk[i];
        if constexpr (GSEG)
            mean_step_v2seg<AVG>(st, ga, pk[0], q[0], mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                 mk(q[1], q[2], q[3]), mk(q[4], q[5], q[6]), bw, ba, s < len);
        else
            mean_step<MODEL, false, AVG>(st, pk[0], q[0], mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                         mk(q[1], q[2], q[3]), mk(q[4], q[5], q[6]), bw, ba, gk, s < len);
#pragma unroll
        for (int i = 0; i < 7; i++) pk[i] = q[i];
    }
    // order-preserving composition over the L lanes of a window; partners sit inside one 16-lane DPP row (L <= 16) or
    // one row further (L = 32: one LDS shuffle level)
#pragma unroll
    for (int stp = 1; stp < L; stp <<= 1) {
        MeanState<false> B;
        GravAcc gB;
        if (stp < 16) {
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) B.R.m[i][j] = row16_down0(st.R.m[i][j], stp);
            B.alpha = mk(row16_down0(st.alpha.x, stp), row16_down0(st.alpha.y, stp), row16_down0(st.alpha.z, stp));
            B.beta = mk(row16_down0(st.beta.x, stp), row16_down0(st.beta.y, stp), row16_down0(st.beta.z, stp));
            B.DT = row16_down0(st.DT, stp);
            if constexpr (GSEG) {
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) { gB.Gam.m[i][j] = row16_down0(ga.Gam.m[i][j], stp); gB.Lam.m[i][j] = row16_down0(ga.Lam.m[i][j], stp); }
            }
        } else {
            B = shfl_down(st, stp);
            if constexpr (GSEG) gB = shfl_down(ga, stp);
        }
        if constexpr (GSEG) grav_combine(ga, st, gB, B);
        mean_combine(st, B);
    }
    if constexpr (GSEG) grav_apply(st, ga, gk);
    if (valid && l == 0) {
        if (A.out.DT) A.out.DT[w] = st.DT;
        if (A.out.alpha) stv3(A.out.alpha + w * 3, st.alpha);
        if (A.out.beta) stv3(A.out.beta + w * 3, st.beta);
        if (A.out.q) {
            const Q4 q = rot_2_quat(st.R);
            double *p = A.out.q + w * 4;
            p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w;
        }
    }
}


// ============================================================================================
// mean kernel, dense layout, WHOLE 128-byte lines: cpi_mean_line_kernel (round 5; CPI_AMD_MEAN_LINE=1)
// ============================================================================================
// "stop paying for half-read lines".  The staged cpi_mean_kernel fetches a window's knots in 112- /
// 168-byte pieces that start anywhere in a 128-byte line; every piece leaves a half-read line behind that the next chunk comes
// back for one chunk period later, and at 8 wavefronts per CU the L2 has evicted a good half of them by then (1.35 x the
// algorithmic bytes on the dense 1 M x 50 launch).  Here a window is fetched LINE BY LINE -- trip j brings line j (TL = 2: lines
// 2 j, 2 j + 1) of each of the wavefront's 64 windows, 16 lanes per line, four whole lines per load instruction -- and the
// doubles go into a per-window RING in LDS at (position in the window) mod RING, from which the window's lane takes its knots
// as they complete.  What makes that cheap is the choice of the 64 windows: the dense layout's window stride is 7 (N + 1)
// doubles, so the offset of a window's first double within its line ("phase") repeats with period P = 16 / gcd(7 (N + 1), 16)
// windows, and a wavefront takes 64 windows of ONE phase class (w, w + P, w + 2 P, ...): every lane's window completes its
// knots at the same trips -- the ring never holds more than 6 + 16 TL doubles, the write slots of a trip are the same for every
// window (one LDS address per lane and trip mod 3 / 5; the element index is an immediate), the consumer's slot is wave-uniform,
// and there is no per-lane bookkeeping at all (158 registers, 12.8 KB of LDS: 12 wavefronts per CU).  The P wavefronts of a
// group of 64 P consecutive windows run on one XCD (block b -> XCD b mod 8) at the same time, so the output lines they share
// are mostly combined in that L2.  Lines are counted from A.knots (no byte outside the caller's array is touched); the launcher
// takes the leading whole groups only when A.knots is 128-byte aligned, never the batch's last 64 P windows (a window's last
// line reaches into the next window).  Bit-identical to cpi_mean_kernel (tests/tools/dma_check.py, and the three-knot parity
// test run against a build that uses it).
// MEASURED (MI355X, profiles/r05_mean_traffic.md; dense 1 M x 50, same box, alternating): HBM traffic 1.35 x -> 1.09 x (reads
// 1.04 x; the strided output stores are written back ~1.8 times) -- and the launch is SLOWER: 677-706 us against 632-662 for the
// shipped three-knot kernel, whatever the lines in flight (one or two trips ahead), the occupancy (9 or 12 wavefronts per CU)
// or the piece (TL = 2: 681-689 us).  Its fetch alone (no arithmetic) takes 642-653 us, its arithmetic alone 384 us.  The
// memory system, not the kernel, sets that: a pure-load probe of the pattern (tools/exp/stride_probe.hip: 64 windows per
// wavefront visited in lock step, B contiguous bytes per window and visit, nothing else) streams 4.6-4.7 TB/s at B = 128, 5.0 at
// 256, 5.6-5.7 at 512, 5.8 at 1024, 6.3 linearly -- DRAM row locality -- and the staged kernel's re-fetched half-lines are served
// by the Infinity Cache beside that stream (tools/exp/mall_probe.hip: a re-touch within 64 MB of traffic costs 0.7 of a DRAM
// fetch and adds to, rather than queues behind, the DRAM stream).  Both designs draw ~4.6-4.7 TB/s from DRAM; B >= 512 at one
// lane per window would need >= 36 KB of LDS per wavefront.  NOT the default: CPI_AMD_MEAN_LINE=1 for A/B measurements only.
#ifndef CPI_MEAN_LINE_TL
#define CPI_MEAN_LINE_TL 1               // lines per window and trip (1 | 2)
#endif
#ifndef CPI_MEAN_LINE_OCC
#define CPI_MEAN_LINE_OCC ((MODEL == 2 || CPI_MEAN_LINE_TL > 1) ? 2 : 3)
#endif
template <int MODEL, bool AVG>
__global__ __launch_bounds__(64, CPI_MEAN_LINE_OCC) void cpi_mean_line_kernel(PreArgs A, int P, long long ngroups) {
    constexpr int TL = CPI_MEAN_LINE_TL;
    constexpr int D = 16 * TL;               // doubles per window and trip = lanes per window of a load instruction = staged elements per lane
    constexpr int WPI = 64 / D;              // windows per load instruction
    // RING: a trip adds D doubles to at most 6 left over (an incomplete knot): 24 slots (12.8 KB per wavefront = 12 per CU) / 40
    constexpr int RING = (D + 6 + 7) / 8 * 8, PITCH = RING + 1;   // odd pitch: the consumer's ds_read_b64 of a half-wave hit 32 distinct even banks
    constexpr int NP = RING / 8;             // write slots repeat every NP trips (D = 16: 3, D = 32: 5)
    __shared__ double tile[64 * PITCH];
    const int lane = threadIdx.x;
    const unsigned b = blockIdx.x, t = b >> 3;
    const int c = (int)(t % (unsigned)P);
    const long long grp = (long long)(t / (unsigned)P) * 8 + (b & 7u);
    if (grp >= ngroups) return;
    const long long S = (long long)(A.N + 1) * 7;          // doubles per window
    const long long wbase = grp * 64 * P + c;               // wave-uniform
    const long long w = wbase + (long long)P * lane;        // this lane's window (as consumer)
    const long long G0 = wbase * S;
    const int ph = (int)(G0 & 15);                          // phase of every window of this wavefront
    const long long PS = (long long)P * S;                  // doubles between consecutive lanes' windows: a multiple of 16
    const double *blk0 = A.knots + (G0 - ph);

    // producer side: element e of lane i = double (i mod D) of the current piece of window WPI e + i / D; position D j + d - ph
    // of the window, slot = position mod RING: NP write addresses per lane (by trip mod NP), the element index is an immediate
    const int g = lane / D, d = lane % D;
    unsigned voff[D];
#pragma unroll
    for (int e = 0; e < D; ++e) voff[e] = (unsigned)(((long long)(WPI * e + g) * PS + d) * 8);
    const int s0 = (d - ph + RING) % RING;
    double *wp[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) wp[k] = tile + g * PITCH + (s0 + k * D) % RING;
    const double *const row = tile + lane * PITCH;

    const V3 bw = ldv3(A.lin + w * 6), ba = ldv3(A.lin + w * 6 + 3);
    V3 gk = mk(0, 0, 0);
    if (MODEL == 2) gk = mul(quat_2_Rot(ldq4(A.qk + w * 4)), mk(A.grav[0], A.grav[1], A.grav[2]));
    MeanState<false> st;
    mean_init(st);

    const int K1 = A.N + 1;                                 // knots per window
    const int T = (int)((ph + S + D - 1) / D);              // trips per window
    double stage[D];
#ifdef CPI_MEAN_LINE_MODE
#pragma unroll
    for (int e = 0; e < D; ++e) stage[e] = 0.01 * (e + 1) + 1e-4 * lane;   // measurement builds (tools/exp/): mode 2 = no fetch
#endif
    auto issue = [&](int j) {
#if defined(CPI_MEAN_LINE_MODE) && CPI_MEAN_LINE_MODE == 2
        (void)j;
#else
        const char *cb = reinterpret_cast<const char *>(blk0) + (long long)j * (D * 8);
#pragma unroll
        for (int e = 0; e < D; ++e) {
            asm volatile("" : "+v"(voff[e]));   // `global_load v, v_off32, s[base]`
            stage[e] = *reinterpret_cast<const double *>(cb + voff[e]);
        }
#endif
    };
    double pk[7] = {0, 0, 0, 0, 0, 0, 0};
    int kdone = 0, jm = 0, p0 = 0;                          // knots integrated; trip mod NP; ring slot of the next knot -- all wave-uniform
    issue(0);
    for (int j = 0; j < T; ++j) {
        double *wb = wp[0];
#pragma unroll
        for (int k = 1; k < NP; ++k) wb = (jm == k) ? wp[k] : wb;
        jm = (jm == NP - 1) ? 0 : jm + 1;
#pragma unroll
        for (int e = 0; e < D; ++e) wb[e * WPI * PITCH] = stage[e];
        __syncthreads();
        if (j + 1 < T) issue(j + 1);
        const int ka = min(K1, (D * (j + 1) - ph) / 7);     // complete knots in the ring or behind it
        for (int kk = kdone; kk < ka; ++kk) {
            double q[7];
            if (p0 <= RING - 7) {
                const double *nk = row + p0;
#pragma unroll
                for (int i = 0; i < 7; i++) q[i] = nk[i];
            } else {
#pragma unroll
                for (int i = 0; i < 7; i++) q[i] = row[(p0 + i) % RING];
            }
            p0 = (p0 + 7 >= RING) ? p0 + 7 - RING : p0 + 7;
#if defined(CPI_MEAN_LINE_MODE) && CPI_MEAN_LINE_MODE == 1
            st.DT += ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + q[6]);    // measurement build: fetch without arithmetic
#else
            if (kk > 0)
                mean_step<MODEL, false, AVG>(st, pk[0], q[0], mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                             mk(q[1], q[2], q[3]), mk(q[4], q[5], q[6]), bw, ba, gk, true);
#endif
#pragma unroll
            for (int i = 0; i < 7; i++) pk[i] = q[i];
        }
        kdone = ka;
        __syncthreads();
    }
    if (A.write_means) {
        if (A.out.DT) A.out.DT[w] = st.DT;
        if (A.out.alpha) stv3(A.out.alpha + w * 3, st.alpha);
        if (A.out.beta) stv3(A.out.beta + w * 3, st.beta);
        if (A.out.q) {
            const Q4 q = rot_2_quat(st.R);
            double *p = A.out.q + w * 4;
            p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w;
        }
    }
}


}  // namespace
