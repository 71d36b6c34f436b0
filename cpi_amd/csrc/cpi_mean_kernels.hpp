// cpi_mean_kernels.hpp -- the mean (+ analytic Jacobian) recursion: cpi_mean_kernel (dense / ragged layouts) and cpi_mean_tiled_kernel (tiled layout).
// Part of the translation unit cpi_mean.hip (included there after cpi_math.hpp / cpi_device_util.hpp; not a stand-alone header).
#pragma once

namespace {

// Number of knots with stamp <= T (stamps non-decreasing).  An IMU stream is sampled almost uniformly, so the answer lies
// within a few knots of the linear interpolation between the stream's ends: gallop from that guess until T is bracketed,
// then bisect the bracket -- typically 3-5 probes inside one or two cache lines, where a plain bisection of a 50 M-knot
// stream takes 26 probes of which the last ten are private to the window (measured: 2.4 KB of extra HBM traffic per window).
__device__ __forceinline__ long long knots_not_after(const double *stream, long long K, double T) {
    const double t0 = stream[0], t1 = stream[(K - 1) * 7];
    if (!(T >= t0)) return 0;
    if (T >= t1) return K;
    long long g = (long long)((T - t0) / (t1 - t0) * (double)(K - 1));
    g = min(max(g, 0ll), K - 1);
    long long lo, hi;                       // invariant: stream[lo - 1] <= T (or lo == 0), stream[hi] > T (or hi == K)
    if (stream[g * 7] <= T) {
        lo = g + 1; hi = K;
        for (long long step = 1; lo < K; step <<= 1) {
            const long long p = min(g + step, K - 1);
            if (stream[p * 7] <= T) { lo = p + 1; if (p == K - 1) break; } else { hi = p; break; }
        }
    } else {
        hi = g; lo = 0;
        for (long long step = 1; hi > 0; step <<= 1) {
            const long long p = max(g - step, 0ll);
            if (stream[p * 7] <= T) { lo = p + 1; break; } else { hi = p; if (p == 0) break; }
        }
    }
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if (stream[mid * 7] <= T) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// The same count with ONE memory round trip in the common case: the four stamps around the interpolation guess are requested
// together (one or two cache lines) and resolved with compares; only a guess that does not bracket T (a stream with gaps)
// falls back to the gallop above.  The answer is unique for non-decreasing stamps, so both routes return the same number.
// Used by the fused cut in the prologue of cpi_mean_kernel<..., CUT = 2>, where every dependent round trip is exposed
// latency of a whole wavefront.  t0 / t1: stamps of the first / last reading.
__device__ __forceinline__ long long knots_not_after_near(const double *stream, long long K, double t0, double t1, double T,
                                                          double &last_stamp) {
    // last_stamp: the stamp of knot (result - 1) when result > 0 -- one of the probes in the common case, so the caller's
    // "stamp of the front reading" costs no further round trip
    last_stamp = t0;
    if (!(T >= t0)) return 0;
    if (T >= t1) { last_stamp = t1; return K; }
    long long g = (long long)((T - t0) / (t1 - t0) * (double)(K - 1));
    g = min(max(g, 1ll), K - 3);          // probes g - 1 .. g + 2 (K >= 4 is checked by the caller)
    const double pa = stream[(g - 1) * 7], pb = stream[g * 7], pc = stream[(g + 1) * 7], pd = stream[(g + 2) * 7];
    if (pa <= T && T < pd) {              // #{t <= T} = (g - 1) + 1 + [pb <= T] + [pc <= T]
        last_stamp = (pc <= T) ? pc : ((pb <= T) ? pb : pa);
        return g + (pb <= T ? 1 : 0) + (pc <= T ? 1 : 0);
    }
    const long long r = knots_not_after(stream, K, T);
    last_stamp = stream[max(r - 1, 0ll) * 7];
    return r;
}

// ============================================================================================
// mean (+ analytic Jacobian) kernel
// ============================================================================================
#ifndef CPI_MEAN_WPS
#define CPI_MEAN_WPS 1
#endif
#ifndef CPI_MEAN_C
#define CPI_MEAN_C 2      // knots per chunk of the staged kernels with several intervals per lane (3 measured: see below)
#endif
#ifndef CPI_MEAN_BIG_C
#define CPI_MEAN_BIG_C 3  // knots per chunk of the BIG instantiations (round 6, short windows: 2 and 5 measured at N = 10 / 20, profiles/r06_short_windows.md)
#endif

// CUT: the windows are cut out of one stream in flight (cpi_preintegrate_stream) -- a template parameter, so that the
// plain-knot instantiations carry none of it (the 10 k-window headline launch is issue-bound: a few extra live registers
// and selects per interval cost it 3-4 %).  1: the cut was made by cpi_cut_windows_kernel (PreArgs::first / count / tstart /
// tend: the workspace route, shared with the covariance kernels); 2: FUSED -- the wavefront cuts its own windows in its
// prologue from PreArgs::update (mean-only requests: no cut kernel, no 56 bytes of workspace traffic per window, and the
// search probes land on the lines the window reads anyway).
// BIG (one lane per window, mean-only, batches that fill the chip many times over): THREE knots per chunk instead of two.
// 21 staged doubles per lane do not fit beside round 3's per-element bookkeeping (a 64-bit pointer, a 32-bit fast-path offset,
// the last valid chunk and the LDS slot per element: 266 registers, one wavefront per SIMD -- which cut the HBM traffic of a 1 M
// launch from 1.26 x to 1.08 x algorithmic but not its time, profiles/r04_mean_chunk_ab.md part B), so BIG keeps ONE 32-bit byte
// offset per element, relative to the wavefront's lowest knot (a wave-uniform base in SGPRs), for the fast path (constant) and
// the per-element path (advanced per chunk) alike, and its LDS tile is flat (pitch 21 doubles: element e of lane i sits at
// 64 e + i, an immediate offset) -- two wavefronts per SIMD again, 2-3 % faster than the two-knot kernel on large batches and 7 %
// on ragged stream windows; the over-fetch at that occupancy is back at 1.37-1.48 x (it is an L2-capacity effect: part D).  The
// launcher admits BIG only where every lane-segment of a wavefront lies within 2^32 bytes above the lowest one: stream windows of
// a stream of < 2^26 readings, or the dense layout (cpi_mean.hip: launch_mean_L).  5 knots per chunk, and two chunks fetched
// back to back per trip, are slower.
template <int MODEL, bool JAC, bool AVG, int L, int CUT, bool BIG = false>
__global__ __launch_bounds__(64, BIG ? 2 : (((MODEL == 2 && !JAC) || (MODEL == 1 && JAC)) && L == 1 ? 2 : CPI_MEAN_WPS)) void cpi_mean_kernel(PreArgs A) {
    static_assert(!BIG || (L == 1 && !JAC), "BIG: one lane per window, mean-only");
    constexpr int WPB = 64 / L;       // windows per wavefront
    // knots staged per lane per chunk: measured on MI355X -- 2 when a lane has several intervals (L <= 8; 20 k x 50 with
    // L = 3: 19.7 -> 18.4 us, 30 k with L = 2: 27.4 -> 24.8 us, 15 k with L = 4: 16.0 -> 15.3 us, 10 k with L = 6:
    // 12.5 -> 11.8 us once the padded second step of an odd last chunk is skipped), 1 when a wave is latency-bound
    // with few intervals per lane (L >= 12: 5 k windows 9.55 vs 9.65 us, 2.5 k 7.7 vs 8.0 us); 3 for BIG (above)
    constexpr int C = BIG ? CPI_MEAN_BIG_C : ((L <= 8 && !JAC) ? CPI_MEAN_C : 1);
    constexpr int SEGD = 7 * C;       // doubles per lane per chunk
    constexpr int PITCH = (SEGD & 1) ? SEGD : SEGD + 1;   // odd pitch (15, 21 doubles): a half-wave's ds_read_b64 hit 32 distinct even banks
    __shared__ double tile[64 * PITCH];
    __shared__ unsigned long long segdesc[64];  // per lane-segment: (first double of the segment << 16) | intervals

    const int lane = threadIdx.x;
    const int grp = lane / L, l = lane - grp * L;
    long long w = (long long)blockIdx.x * WPB + grp;
    const bool valid = (w < A.W) && (grp < WPB);   // L not a power of two leaves 64 - WPB*L idle lanes
    if (grp >= WPB) w = (long long)blockIdx.x * WPB;   // idle lanes shadow the block's first window (stays near the block)
    if (w >= A.W) w = A.W - 1;
    constexpr bool cut = CUT != 0;
    int n;
    long long k0;
    // Windows cut out of a stream in flight: the window's first knot takes the stamp t_start, and a partial tail interval
    // has NO knot in memory -- it is the last real knot's reading held until t_end.  The lane that owns the tail fetches one
    // knot less and builds that knot from its predecessor when it gets there.
    double t_start = 0.0, t_end = 0.0;
    bool tail = false;
    if constexpr (CUT == 2) {
        // the arithmetic of cpi_cut_windows_kernel (GraphSolver_IMU.cpp:50-69 as a closed form), per lane, in registers
        const double ts0 = A.knots[0], ts1 = A.knots[(A.K - 1) * 7];
        const double T = A.update[w], Tp = A.update[w > 0 ? w - 1 : 0];
        double stT, stP;
        const long long cT = knots_not_after_near(A.knots, A.K, ts0, ts1, T, stT);
        const long long cP = knots_not_after_near(A.knots, A.K, ts0, ts1, Tp, stP);
        const long long fp = (w > 0) ? max(cP - 1, 0ll) : 0ll;
        t_start = (w > 0) ? fmax(Tp, ts0) : ts0;
        const long long fu = max(max(cT - 1, 0ll), fp);
        const int m = (int)min(fu - fp, (long long)0x3fffffff);
        const double front_t = (m > 0) ? stT : t_start;                  // m > 0: fu = cT - 1 > 0, whose stamp the search returned
        const bool tl = (T - front_t) > 0;
        const int cnt = m + (tl ? 1 : 0);
        if (valid && l == 0) A.count_out[w] = cnt;                       // the TRUE count (cpi_stream_counts)
        k0 = fp;
        n = min(cnt, A.N);
        t_end = T;
        tail = tl && cnt <= A.N;                                         // a truncated window has lost its tail
    } else {
        n = A.count ? min(max(A.count[w], 0), A.N) : A.N;   // a count outside [0, N] must not corrupt the packed descriptors
        k0 = A.first ? A.first[w] : w * (long long)(A.N + 1);
        if constexpr (CUT == 1) {
            t_start = A.tstart[w]; t_end = A.tend[w];
            tail = (t_end == t_end) && (A.count[w] <= A.N);              // NaN = no tail; a truncated window has lost it
        }
    }
    const int per = (n + L - 1) / L;
    const int s0 = min(n, l * per), s1 = min(n, s0 + per);
    const int len = s1 - s0;
    const int maxlen = __builtin_amdgcn_readfirstlane(wave_max(len));   // wave-uniform: loop control stays scalar
    const bool tailseg = cut && tail && (s1 == n) && (len > 0);
    const int len_f = len - (tailseg ? 1 : 0);                          // knots after the segment's first that exist in memory
    // First knot of the segment IN MEMORY.  Knot s0 always exists (a window owns count + 1 knots) -- except the virtual tail
    // knot, which only an EMPTY trailing segment (s0 == n, lanes beyond ceil(n / per)) can start on: nothing of such a segment is
    // ever consumed, but its first knot is still fetched (pk below, and the staging path re-reads a never-valid element's base
    // knot), and when the window ends on the stream's last reading (update time past the last stamp) knot k0 + n lies 56 bytes
    // behind the caller's buffer -- unmapped memory, or NaN bits that reach the state through 0 * NaN on the dt = 0 steps with
    // imu_avg.  Such a segment is based on the last real knot instead.
    const int sb = (cut && tail && s0 == n && n > 0) ? s0 - 1 : s0;

    // (Deriving the descriptors of a dense layout arithmetically instead of through LDS was measured: +0.35 us per
    // 13 us launch -- the 64-bit integer arithmetic costs more than the shuffle reduction and the LDS round trip.)
    segdesc[lane] = ((unsigned long long)((k0 + sb) * 7) << 16) | (unsigned long long)(unsigned)len_f;

    const V3 bw = ldv3(A.lin + w * 6), ba = ldv3(A.lin + w * 6 + 3);
    V3 gk = mk(0, 0, 0);
    if (MODEL == 2) gk = mul(quat_2_Rot(ldq4(A.qk + w * 4)), mk(A.grav[0], A.grav[1], A.grav[2]));

    double pk[7];
    {
        const double *kb = A.knots + (k0 + sb) * 7;
#pragma unroll
        for (int i = 0; i < 7; i++) pk[i] = kb[i];
        if (cut && s0 == 0) pk[0] = t_start;
    }
    MeanState<JAC> st;
    mean_init(st);
    // model 2, mean-only, several lanes per window: a lane integrates its segment from the raw specific force and
    // accumulates the segment's gravity response (cpi_math.hpp: mean_step_v2seg); gravity is applied after the tree
    constexpr bool GSEG = (MODEL == 2) && !JAC && (L > 1);
    GravAcc ga;
    if (GSEG) grav_init(ga);
    __syncthreads();

    // Analytic-Jacobian variant of model 1, one lane per window (large batches): the recursion is bound by registers
    // (61 doubles of state + the per-interval 3x3 temporaries), not by HBM, so it streams its knots straight into
    // registers, one interval ahead, instead of through the coalescing LDS stage -- that frees the stage's address /
    // staging registers and lets two wavefronts share a SIMD (256 registers + 36 B of scratch each).  Measured inside
    // "V1 full" (covariance kernel + this one): 1.405 -> 1.376 ms per 100 k windows, 13.25 -> 13.10 ms per 1 M.  With
    // several lanes per window (small, latency-bound batches) it loses (10 k windows: 192 -> 205 us), so those keep the stage.
    constexpr bool DIRECT = JAC && (MODEL == 1) && (L == 1);
    if constexpr (DIRECT) {
        const double *kp = A.knots + (k0 + s0) * 7;
        double nx[7];
        {
            const double *kb = kp + 7 * min(1, len_f);
#pragma unroll
            for (int i = 0; i < 7; i++) nx[i] = kb[i];
        }
        for (int sidx = 0; sidx < maxlen; ++sidx) {
            double q[7];
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = nx[i];
            {
                const double *kb = kp + 7 * min(sidx + 2, len_f);   // knot s0 + len_f is the segment's last one in memory: always valid
#pragma unroll
                for (int i = 0; i < 7; i++) nx[i] = kb[i];
            }
            if constexpr (cut) {      // the tail knot: the predecessor's reading under the update time
                const bool here = tailseg && sidx == len - 1;
                q[0] = here ? t_end : q[0];
#pragma unroll
                for (int i = 1; i < 7; i++) q[i] = here ? pk[i] : q[i];
            }
            mean_step<MODEL, JAC, AVG>(st, pk[0], q[0], mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                       mk(q[1], q[2], q[3]), mk(q[4], q[5], q[6]), bw, ba, gk, sidx < len);
#pragma unroll
            for (int i = 0; i < 7; i++) pk[i] = q[i];
        }
    } else {
    // Tile element idx = e*64 + lane belongs to segment idx / SEGD at offset idx % SEGD, so consecutive
    // lanes read consecutive doubles of (mostly) one segment: coalesced.  Everything that does not depend
    // on the chunk index is hoisted: per staged element a lane keeps one pointer and the last chunk for
    // which its knot exists (later chunks re-read that knot; the value is never consumed), so the hot loop
    // spends ~3 VALU per element on addressing and no load is ever out of bounds.
    double stage[SEGD];
    const double *sptr[SEGD];   // !BIG
    unsigned voff[SEGD];   // byte offset of the element from blk0: the fast path's constant (dense layouts, uniform streams); BIG: of both paths
    int smax[SEGD];
    int tofs[SEGD];             // !BIG (BIG: the tile is flat, element e of lane i at 64 e + i)
    const double *blk0 = A.knots + (long long)blockIdx.x * WPB * (long long)(A.N + 1) * 7;   // wave-uniform (dense layout)
    bool fast_stream = false;
    const long long b = k0 + sb;
    long long b0 = 0;
    if constexpr (BIG) {
        // the wavefront's lowest first knot (stream windows out of time order may start below lane 0's)
        const long long bf = readfirstlane64(b);
        b0 = bf - (long long)wave_max((int)min(max(bf - b, 0ll), 0x7fffffffll));
        blk0 = A.knots + b0 * 7;
    }
    if constexpr (cut) {
        // The stream entry's twin of the dense layout's fast path below: "wave-uniform base + chunk stride in SGPRs + constant
        // 32-bit lane offsets" is valid for a stream whenever (a) every lane-segment of the wavefront has the same length --
        // then no lane ever CONSUMES a knot behind its own segment (the padded step of an odd last chunk is skipped, a tail
        // knot is replaced by a select), so reading on is harmless whatever those knots hold --, (b) the segments lie within
        // 2^30 bytes above the first one and (c) the furthest read stays inside the stream (PreArgs::K).  A uniform update
        // grid satisfies all three for every wavefront but the last; ragged wavefronts keep the per-element path.
        if constexpr (!BIG) b0 = readfirstlane64(b);
        const int nch = (maxlen + C - 1) / C;
        const bool ok = (A.K > 0) && (len == maxlen) && (b >= b0) && (b - b0 < (1ll << 24)) && (b + (long long)nch * C <= A.K - 1);
        fast_stream = __all(ok);
        if (fast_stream) blk0 = A.knots + b0 * 7;
    }
    {
        // (Issuing all SEGD descriptor reads before using the first -- one LDS round trip instead of SEGD dependent ones,
        // which hipcc keeps in program order with an s_waitcnt after each -- was measured: 12.55 vs 12.33 us per launch
        // at 10 k windows, i.e. slower; the wavefronts wait for the first HBM burst either way and start less staggered.)
        int seg = lane / SEGD, off = lane - seg * SEGD;
#pragma unroll
        for (int e = 0; e < SEGD; ++e) {
            const unsigned long long d = segdesc[seg];
            const long long base = (long long)(d >> 16);
            const int slen = (int)(d & 0xffffULL);
            const int kn = off / 7;                       // knot (1 + kn) of chunk 0
            const bool ok = slen >= 1 + kn;
            if constexpr (BIG) {
                voff[e] = (unsigned)((base - b0 * 7 + (ok ? 7 + off : off - 7 * kn)) * 8);   // < 2^32: the launcher's admission rule
            } else {
                sptr[e] = A.knots + base + (ok ? 7 + off : off - 7 * kn);
                voff[e] = (unsigned)((sptr[e] - blk0) * 8);   // only used when safe_overread (then 0 <= offset < 2^32)
                tofs[e] = seg * PITCH + off;
            }
            smax[e] = ok ? (slen - 1 - kn) / C : 0;       // never-valid elements keep re-reading knot 0
            off += 64 % SEGD; seg += 64 / SEGD;   // idx advances by 64 per staged element
            if (off >= SEGD) { off -= SEGD; seg += 1; }
        }
    }
    // Dense layout, not one of the last waves: reading a few knots past a short segment's end stays inside
    // the knot array, so every chunk is "block base + chunk stride (scalar) + constant lane offset".
    // (Not with per-window counts: the knots behind a short window's last interval belong to the caller's dense array and
    // may never have been written -- a NaN there would reach the state through 0 * NaN on the inactive steps.  The
    // per-element path below stops at the segment's end and re-reads its last, valid knot instead.)
    const bool safe_overread = cut ? fast_stream
                                   : ((A.first == nullptr) && (A.count == nullptr) && ((long long)(blockIdx.x + 1) * WPB + 2 < A.W));
    auto issue = [&](int it) {
        if (safe_overread) {
            // scalar base (advanced by SALU) + constant 32-bit lane offsets: no vector arithmetic per element
            const char *cb = reinterpret_cast<const char *>(blk0) + (long long)it * (SEGD * 8);
#pragma unroll
            for (int e = 0; e < SEGD; ++e) {
                asm volatile("" : "+v"(voff[e]));   // keeps the zero-extension next to the load: `global_load v, v_off32, s[base]`
                stage[e] = *reinterpret_cast<const double *>(cb + voff[e]);
            }
        } else if constexpr (BIG) {
            const char *cb = reinterpret_cast<const char *>(blk0);
#pragma unroll
            for (int e = 0; e < SEGD; ++e) {
                asm volatile("" : "+v"(voff[e]));
                stage[e] = *reinterpret_cast<const double *>(cb + voff[e]);
                voff[e] += (it < smax[e]) ? (unsigned)(SEGD * 8) : 0u;
            }
        } else {
#pragma unroll
            for (int e = 0; e < SEGD; ++e) { stage[e] = *sptr[e]; sptr[e] += (it < smax[e]) ? SEGD : 0; }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int e = 0; e < SEGD; ++e) {
            if constexpr (BIG) tile[e * 64 + lane] = stage[e]; else tile[tofs[e]] = stage[e];
        }
    };

    // One chunk ahead: the HBM round trip of chunk it+1 overlaps the FP64 work of chunk it.  Measured alternatives:
    // a TRUE two-chunk pipeline (two register stages, every path issuing the same loads so that hipcc emits the partial
    // wait s_waitcnt vmcnt(14) -- one conditional issue in the loop and it drains the queue with vmcnt(0)) is 8 % slower
    // at 10 k windows x 50 (13.5 vs 12.5 us: the first chunk's data queues behind the second's) and 5 % slower at 1 M;
    // a double-buffered LDS tile with the next chunk read back into registers during the integration: +2 %.
    // Per-wavefront time stamps explain why: with 1000 wavefronts in flight a chunk is 3.6 MB and takes 0.89 us
    // (0.74 us with 625 wavefronts, 1.2 us with 2000) -- the loop streams at ~4 TB/s and is paced by the memory
    // system, not by the latency of one wavefront's accesses.
    // Stream windows on a uniform update grid: EVERY lane-segment of the wavefront ends in its window's tail interval and all
    // are equally long -- the tail step is then peeled off behind the loop and the loop carries no per-step selects (14
    // v_cndmask per interval of ~300 VALU; measured on the 1 M x 51 stream: 724 -> see DESIGN.md 3.1b).  Wave-uniform.
    const bool utail = cut && __all(tailseg && len == maxlen);
    const int nsteps = utail ? maxlen - 1 : maxlen;
    const int nchunks = (nsteps + C - 1) / C;
    if (nchunks > 0) issue(0);
    for (int it = 0; it < nchunks; ++it) {
        commit();
        __syncthreads();
        if (it + 1 < nchunks) issue(it + 1);
#pragma unroll   // C <= 2: the two steps of a chunk share one basic block (no knot copy between them)
        for (int c = 0; c < C; ++c) {
            const int s = it * C + c;
            if (C > 1 && s >= nsteps) break;   // wave-uniform: no lane has this interval (odd longest segment)
            const double *nk = &tile[lane * PITCH + c * 7];
            double q[7];
#pragma unroll
            for (int i = 0; i < 7; i++) q[i] = nk[i];
            if constexpr (cut) {      // the tail knot: the predecessor's reading under the update time
                if (!utail) {
                    const bool here = tailseg && s == len - 1;
                    q[0] = here ? t_end : q[0];
#pragma unroll
                    for (int i = 1; i < 7; i++) q[i] = here ? pk[i] : q[i];
                }
            }
            if constexpr (GSEG)
                mean_step_v2seg<AVG>(st, ga, pk[0], q[0], mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                     mk(q[1], q[2], q[3]), mk(q[4], q[5], q[6]), bw, ba, s < len);
            else
                mean_step<MODEL, JAC, AVG>(st, pk[0], q[0], mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                           mk(q[1], q[2], q[3]), mk(q[4], q[5], q[6]), bw, ba, gk, s < len);
#pragma unroll
            for (int i = 0; i < 7; i++) pk[i] = q[i];
        }
        __syncthreads();
    }
    if constexpr (cut) {
        if (utail) {   // the peeled tail interval [stamp of the last real knot, t_end], that knot's reading held
            if constexpr (GSEG)
                mean_step_v2seg<AVG>(st, ga, pk[0], t_end, mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                     mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]), bw, ba, true);
            else
                mean_step<MODEL, JAC, AVG>(st, pk[0], t_end, mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]),
                                           mk(pk[1], pk[2], pk[3]), mk(pk[4], pk[5], pk[6]), bw, ba, gk, true);
        }
    }

    }   // !DIRECT

    // order-preserving composition tree over the L lanes of a window (earlier = lower lane)
#pragma unroll
    for (int stp = 1; stp < L; stp <<= 1) {
        MeanState<JAC> B = shfl_down(st, stp);
        GravAcc gB;
        if constexpr (GSEG) gB = shfl_down(ga, stp);
        if ((L & (L - 1)) != 0) {
            // L not a power of two: lane l + stp may belong to the next window -- compose with the identity instead
            if (l + stp >= L) { mean_init(B); if (GSEG) grav_init(gB); }
        }
        if constexpr (GSEG) grav_combine(ga, st, gB, B);   // needs st.R / B.DT before they are composed
        mean_combine(st, B);
    }
    if constexpr (GSEG) grav_apply(st, ga, gk);

    if (valid && l == 0) {
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
        if (JAC && A.write_jac) {
            if (A.out.J_q) stm3_cm(A.out.J_q + w * 9, st.Jq);
            if (A.out.J_a) stm3_cm(A.out.J_a + w * 9, st.Ja);
            if (A.out.J_b) stm3_cm(A.out.J_b + w * 9, st.Jb);
            if (A.out.H_a) stm3_cm(A.out.H_a + w * 9, st.Ha);
            if (A.out.H_b) stm3_cm(A.out.H_b + w * 9, st.Hb);
            if (MODEL == 2) {
                if (A.out.O_a) stm3_cm(A.out.O_a + w * 9, st.Oa);
                if (A.out.O_b) stm3_cm(A.out.O_b + w * 9, st.Ob);
            }
        }
    }
}

// ============================================================================================
// mean kernel on the TILED layout: knots of 64 windows interleaved per step (cpi_preintegrate_tiled_batch)
// ============================================================================================
// tiles[b][s][k][i] = field k (t, w, a) of knot s of window 64 b + i.  A wavefront owns tile b, lane i window 64 b + i, and
// step s reads seven fully coalesced 512-byte rows: the whole batch is ONE linear stream per wavefront, every byte
// fetched once, no LDS, no staging, ~100 registers (4 wavefronts per SIMD) -- the layout the recursion wants on this
// memory system, for producers that can write it (a batch assembler that places knot s of window w at its tile slot instead of
// at w (N+1) + s costs nothing extra).  Knots are prefetched three steps ahead in registers.
// (TiledArgs: cpi_args.hpp)
#ifndef CPI_TILED_OCC
#define CPI_TILED_OCC (MODEL == 2 ? 2 : 3)
#endif
#ifndef CPI_TILED_BUFS
#define CPI_TILED_BUFS 5
#endif

// SPLIT (small batches: fewer tiles than the chip has SIMDs): a workgroup of S = blockDim.x / 64 wavefronts owns the
// tile; wavefront j integrates the steps [j per, (j + 1) per) of all 64 windows from the identity (model 2: from the raw
// specific force, with the segment's gravity response -- cpi_math.hpp mean_step_v2seg), parks its segment in LDS, and
// wavefront 0 composes the S segments in order (mean_combine / grav_combine: the composition cpi_mean_kernel uses
// across the lanes of a window).  Each wavefront still reads one linear stream.
template <int MODEL, bool AVG, bool COUNTED, bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 512 : 64, SPLIT ? 1 : CPI_TILED_OCC) void cpi_mean_tiled_kernel(TiledArgs A) {
    constexpr bool GSEG = SPLIT && MODEL == 2;
    constexpr int NF = GSEG ? 34 : 16;            // doubles of a parked segment
    extern __shared__ double seg[];               // [S - 1][NF][64]
    const int lane = threadIdx.x & 63;
    const int j = SPLIT ? (int)(threadIdx.x >> 6) : 0, S = SPLIT ? (int)(blockDim.x >> 6) : 1;
    const long long w = (long long)blockIdx.x * 64 + lane;
    const bool valid = w < A.W;
    const long long wc = valid ? w : A.W - 1;
    const int n = valid ? (COUNTED ? min(max(A.count[wc], 0), A.N) : A.N) : 0;
    const int nmax = COUNTED ? __builtin_amdgcn_readfirstlane(wave_max(n)) : A.N;
    const int per = SPLIT ? (A.N + S - 1) / S : A.N;
    const int sb = __builtin_amdgcn_readfirstlane(j * per), se = min(sb + per, nmax);   // this wavefront's steps
    const double *tb = A.tiles + (long long)blockIdx.x * A.ts + lane;   // a step of a tile: 448 doubles = 7 fields x 64 windows
    const V3 bw = ldv3(A.lin + wc * 6), ba = ldv3(A.lin + wc * 6 + 3);
    V3 gk = mk(0, 0, 0);
    if (MODEL == 2 && j == 0) gk = mul(quat_2_Rot(ldq4(A.qk + wc * 4)), mk(A.grav[0], A.grav[1], A.grav[2]));
    auto load = [&](double (&k)[7], int s) {
        // COUNTED: past its own last knot a lane re-reads that knot (dt = 0) -- what lies behind it in the column is
        // never read.  Otherwise the row offset is wave-uniform (scalar address arithmetic).
        const double *p = tb + (long long)(COUNTED ? min(s, n) : min(s, A.N)) * A.ss;
        // non-temporal: every byte of a tile is read exactly once, by one wavefront (round 4, same-box A/B: 1 M x 50 582-588 ->
        // 565-569 us, model 2 561-576 -> 540 us, 100 k 60.5 -> 56 us; the same hint on the dense layout's staged loads, whose
        // 128-byte lines ARE touched again by the next chunk, costs 35 %: 619 -> 840-874 us)
#pragma unroll
        for (int f = 0; f < 7; f++) k[f] = __builtin_nontemporal_load(p + f * 64);
    };
    MeanState<false> st;
    mean_init(st);
    GravAcc ga;
    if (GSEG) grav_init(ga);
    // the knot buffers rotate by NAME over one unrolled trip (a rolled loop spends 28 v_mov_b64 per step on it)
#define CPI_TSTEP(a, b, e, S_)                                                                                        \
    load(e, (S_) + CPI_TILED_BUFS - 1);                                                                               \
    if constexpr (GSEG)                                                                                               \
        mean_step_v2seg<AVG>(st, ga, a[0], b[0], mk(a[1], a[2], a[3]), mk(a[4], a[5], a[6]), mk(b[1], b[2], b[3]),    \
                             mk(b[4], b[5], b[6]), bw, ba, (S_) < n);                                                 \
    else                                                                                                              \
        mean_step<MODEL, false, AVG>(st, a[0], b[0], mk(a[1], a[2], a[3]), mk(a[4], a[5], a[6]), mk(b[1], b[2], b[3]), \
                                     mk(b[4], b[5], b[6]), bw, ba, gk, (S_) < n)
#if CPI_TILED_BUFS == 5
    double k0[7], k1[7], k2[7], k3[7], k4[7];
    load(k0, sb); load(k1, sb + 1); load(k2, sb + 2); load(k3, sb + 3);
    for (int s = sb; s < se; s += 5) {
        CPI_TSTEP(k0, k1, k4, s);
        if (s + 1 >= se) break;
        CPI_TSTEP(k1, k2, k0, s + 1);
        if (s + 2 >= se) break;
        CPI_TSTEP(k2, k3, k1, s + 2);
        if (s + 3 >= se) break;
        CPI_TSTEP(k3, k4, k2, s + 3);
        if (s + 4 >= se) break;
        CPI_TSTEP(k4, k0, k3, s + 4);
    }
#elif CPI_TILED_BUFS == 4
    double k0[7], k1[7], k2[7], k3[7];
    load(k0, sb); load(k1, sb + 1); load(k2, sb + 2);
    for (int s = sb; s < se; s += 4) {
        CPI_TSTEP(k0, k1, k3, s);
        if (s + 1 >= se) break;
        CPI_TSTEP(k1, k2, k0, s + 1);
        if (s + 2 >= se) break;
        CPI_TSTEP(k2, k3, k1, s + 2);
        if (s + 3 >= se) break;
        CPI_TSTEP(k3, k0, k2, s + 3);
    }
#else
    double k0[7], k1[7], k2[7];
    load(k0, sb); load(k1, sb + 1);
    for (int s = sb; s < se; s += 3) {
        CPI_TSTEP(k0, k1, k2, s);
        if (s + 1 >= se) break;
        CPI_TSTEP(k1, k2, k0, s + 1);
        if (s + 2 >= se) break;
        CPI_TSTEP(k2, k0, k1, s + 2);
    }
#endif
#undef CPI_TSTEP
    if constexpr (SPLIT) {
        auto park = [&](int f, double v) { seg[((j - 1) * NF + f) * 64 + lane] = v; };
        if (j > 0) {
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) park(r * 3 + c, st.R.m[r][c]);
            park(9, st.alpha.x); park(10, st.alpha.y); park(11, st.alpha.z);
            park(12, st.beta.x); park(13, st.beta.y); park(14, st.beta.z); park(15, st.DT);
            if constexpr (GSEG) {
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) { park(16 + r * 3 + c, ga.Gam.m[r][c]); park(25 + r * 3 + c, ga.Lam.m[r][c]); }
            }
        }
        __syncthreads();
        if (j > 0) return;
        for (int jj = 1; jj < S; ++jj) {        // earlier o later, in order
            const double *sp = seg + (jj - 1) * NF * 64 + lane;
            MeanState<false> B;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) B.R.m[r][c] = sp[(r * 3 + c) * 64];
            B.alpha = mk(sp[9 * 64], sp[10 * 64], sp[11 * 64]);
            B.beta = mk(sp[12 * 64], sp[13 * 64], sp[14 * 64]);
            B.DT = sp[15 * 64];
            if constexpr (GSEG) {
                GravAcc gB;
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) { gB.Gam.m[r][c] = sp[(16 + r * 3 + c) * 64]; gB.Lam.m[r][c] = sp[(25 + r * 3 + c) * 64]; }
                grav_combine(ga, st, gB, B);    // before mean_combine: needs st.R and B.DT as they are
            }
            mean_combine(st, B);
        }
        if constexpr (GSEG) grav_apply(st, ga, gk);
    }
    if (!valid) return;
    if (A.out.DT) A.out.DT[w] = st.DT;
    if (A.out.alpha) stv3(A.out.alpha + w * 3, st.alpha);
    if (A.out.beta) stv3(A.out.beta + w * 3, st.beta);
    if (A.out.q) {
        const Q4 q = rot_2_quat(st.R);
        double *p = A.out.q + w * 4;
        p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w;
    }
}
#ifdef CPI_EXPERIMENTS
// measurement only (CPI_AMD_BLK_MODE=1): the tiled stream alone -- the same loads, one add per value
__global__ __launch_bounds__(64, 3) void cpi_tiled_fetch_probe_kernel(TiledArgs A) {
    const int lane = threadIdx.x;
    const long long w = (long long)blockIdx.x * 64 + lane;
    const double *tb = A.tiles + (long long)blockIdx.x * A.ts + lane;
    auto load = [&](double (&k)[7], int s) {
        const double *p = tb + (long long)min(s, A.N) * A.ss;
#pragma unroll
        for (int f = 0; f < 7; f++) k[f] = p[f * 64];
    };
    double k0[7], k1[7], k2[7], k3[7], acc = 0;
    load(k0, 0); load(k1, 1); load(k2, 2); load(k3, 3);
    for (int s = 0; s < A.N; ++s) {
        double k4[7];
        load(k4, s + 4);
#pragma unroll
        for (int f = 0; f < 7; f++) { acc += k0[f]; k0[f] = k1[f]; k1[f] = k2[f]; k2[f] = k3[f]; k3[f] = k4[f]; }
    }
    if (w < A.W && A.out.DT) A.out.DT[w] = acc;
}
#endif
// knots[W][N+1][7] (first == NULL) or a shared stream indexed by first[W] / count[W] (the ragged layout of
// cpi_preintegrate_batch) -> tiles[ceil(W/64)][N+1][7][64].  Rows past a window's last knot repeat that knot and columns
// past W repeat window W - 1: finite padding the kernels never integrate.
__global__ __launch_bounds__(256) void cpi_tile_knots_kernel(long long W, int N, const double *knots, const long long *first,
                                                             const int *count, double *tiles, long long ts, long long ss) {
    const long long total = ((W + 63) / 64) * (long long)(N + 1) * 448;
    for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long long)gridDim.x * 256) {
        const int i = (int)(o & 63);
        const long long r = o >> 6;
        const int f = (int)(r % 7);
        const long long bs = r / 7;
        const int sidx = (int)(bs % (N + 1));
        const long long b = bs / (N + 1);
        const long long w = min(b * 64 + i, W - 1);
        const long long k0 = first ? first[w] : w * (long long)(N + 1);
        const int n = count ? min(max(count[w], 0), N) : N;
        tiles[b * ts + sidx * ss + f * 64 + i] = knots[(k0 + min(sidx, n)) * 7 + f];
    }
}

// Window assembly on the device, straight into the tiled layout (cpi_assemble_tiles): ONE IMU stream cut at successive
// update times with the semantics of GraphSolver::createimufactor_cpi_v1/v2 (GraphSolver_IMU.cpp:50-69) -- whole
// intervals while imu_times[1] <= updatetime, then the partial tail interval with the front reading repeated, after which
// the front stamp is overwritten by the update time.  The reference walks a deque, i.e. window u starts where window
// u - 1 stopped; for a stream whose stamps are non-decreasing (and update times likewise) that state is a pure function of
// the previous update time, so every window is independent here:
//     front(T)  = max(#{knots with t <= T} - 1, 0)          (the deque's front index after the window ending at T)
//     stamp(T)  = max(T, t_0)                               (the front stamp after that window)
// One wavefront per tile, one lane per window; a row of the tile is seven coalesced 512-byte stores.
// The same cut WITHOUT moving a knot (cpi_preintegrate_stream): per window the front reading's index, the interval count
// (whole + tail), the start stamp and the update time of a tail interval (NaN: none) -- 28 bytes per window; the
// preintegration kernels then read the stream in place (PreArgs::tstart / tend).
__global__ __launch_bounds__(256) void cpi_cut_windows_kernel(long long K, const double *stream, long long U, const double *update,
                                                              long long *first, int *count, double *tstart, double *tend) {
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    if (u >= U) return;
    const double t0 = stream[0], t1 = stream[(K - 1) * 7], T = update[u];
    const bool near_ok = K >= 4;          // one round trip per search in the common case (knots_not_after_near)
    long long fp = 0;
    double start_t = t0, stT = t0, stP;
    const long long cT = near_ok ? knots_not_after_near(stream, K, t0, t1, T, stT) : knots_not_after(stream, K, T);
    if (u > 0) {
        const double Tp = update[u - 1];
        fp = max((near_ok ? knots_not_after_near(stream, K, t0, t1, Tp, stP) : knots_not_after(stream, K, Tp)) - 1, 0ll);
        start_t = fmax(Tp, t0);
    }
    const long long fu = max(max(cT - 1, 0ll), fp);
    const int m = (int)min(fu - fp, (long long)0x3fffffff);
    const double front_t = (m > 0) ? (near_ok ? stT : stream[fu * 7]) : start_t;
    const bool tail = (T - front_t) > 0;
    first[u] = fp; count[u] = m + (tail ? 1 : 0); tstart[u] = start_t;
    tend[u] = tail ? T : __builtin_nan("");
}
// Rows leave as coalesced 512-byte stores (lane = window).  It is a copy with a transposition in it; designs measured on the
// 1 M x 50 batch (profiles/r04_assembler.md):
//   * round 3: lane i loads its own window's knots (8 B per lane and instruction, 64 different 128-byte lines per instruction):
//     1.46-1.48 ms = 3.9 TB/s of read + write;
//   * SHIPPED: the RB = 8 rows of a trip travel as LDS-DMA (glds16) -- one instruction fetches the 448 contiguous bytes of TWO
//     windows (28 lanes x 16 B each), 32 instructions fill a 33-KB image of the trip's 64 x 8 knots, no staging registers, 8
//     lines per instruction instead of 64 --, and lane i then reads ITS window's values out of the image (pitch 1040 B per
//     instruction image: 2-way bank conflicts, the minimum of this placement) for the row stores: 1.16-1.21 ms = 4.7-4.9 TB/s
//     algorithmic, 5.96 TB/s of real traffic (the 8-byte aligned pieces over-fetch 1.49 x on the read side);
//   * 16-byte stores / 4 rows per trip (9 wavefronts per CU): 1.20 / 1.27 ms -- neither store issue nor occupancy paces it;
//   * every line once: a wavefront owning 16 whole windows (128-byte row pieces: 1.79 ms -- small write pieces stream at half
//     rate), line-aligned 512-byte trips with a carry for the straddling knot, rows stored out of step (reads 1.14 x, writes
//     + 29 %: 1.40 ms) or in step with a 320-byte carry (traffic 1.05 x in total, but 54 KB of LDS = 3 wavefronts per CU:
//     1.34 ms).  Commit "experiment (not shipped): line-aligned assembler" holds the last of them.
//   * round 5: the next trip's LDS-DMA issued under the current trip's row stores (4 rows per trip, two 17-KB images, the landing
//     waited for as vmcnt(28) on exact store counts): reads 1.36 x -> 1.16 x, latency hidden, 1.13-1.16 ms -- the SAME time
//     (profiles/r05_assembler.md; commit "experiment (not shipped): assembler with the next trip's DMA under the row stores").
//     5.0 TB/s of algorithmic read + write is what a 1 : 1 copy of 448-byte pieces into 512-byte rows gets on this memory system.
// The DMA route needs the wavefront's windows within 2^24 knots above the first one and K >= RB (wave-uniform test); any
// other wavefront takes the per-lane loads of round 3.
#ifndef CPI_ASM_RB
#define CPI_ASM_RB 8
#endif
__global__ __launch_bounds__(64) void cpi_assemble_tiles_kernel(AssembleArgs A) {
    constexpr int RB = CPI_ASM_RB;              // rows per trip
    constexpr int PPW = RB * 56 / 16;           // 16-byte pieces of a window's trip: 28
    constexpr int WPI = 64 / PPW;               // windows per DMA instruction: 2
    constexpr int NI = 64 / WPI;                // DMA instructions per trip: 32
    constexpr int IMG = (RB == 8) ? 1040 : 1072;   // bytes between instruction images: 2-way (RB = 8) / 3-way (RB = 4) read conflicts, the minima of these placements
    static_assert((RB == 8 && WPI == 2) || (RB == 4 && WPI == 4), "RB = 8: two windows per instruction; RB = 4: four");
    __shared__ __attribute__((aligned(1024))) char img[NI * IMG];
    __shared__ int srel[64];
    const int lane = threadIdx.x;
    const long long u = (long long)blockIdx.x * 64 + lane;
    const bool valid = u < A.U;
    const long long uc = valid ? u : A.U - 1;
    const double t0 = A.stream[0], tl_ = A.stream[(A.K - 1) * 7];
    const double T = A.update[uc];
    long long fp = 0;
    double start_t = t0;
    double stT, stP;
    const bool near_ok = A.K >= 4;
    const long long cT = near_ok ? knots_not_after_near(A.stream, A.K, t0, tl_, T, stT) : knots_not_after(A.stream, A.K, T);
    if (uc > 0) {
        const double Tp = A.update[uc - 1];
        const long long cP = near_ok ? knots_not_after_near(A.stream, A.K, t0, tl_, Tp, stP) : knots_not_after(A.stream, A.K, Tp);
        fp = max(cP - 1, 0ll);
        start_t = fmax(Tp, t0);
    }
    const long long fu = max(max(cT - 1, 0ll), fp);
    const int m = (int)min(fu - fp, (long long)0x3fffffff);          // whole intervals
    const double front_t = (m > 0) ? (near_ok ? stT : A.stream[fu * 7]) : start_t;   // m > 0: fu = cT - 1, whose stamp the search returned
    const bool tail = (T - front_t) > 0;
    const int cnt = m + (tail ? 1 : 0);
    if (valid) A.count[u] = cnt;                                      // the TRUE count: a caller can check max(count) <= N
    const int rows = min(cnt, A.N);                                   // rows 0 .. rows exist in the tile
    const int rmax = __builtin_amdgcn_readfirstlane(wave_max(rows));
    double *tb = A.tiles + (long long)blockIdx.x * A.ts + lane;

    const long long base = min(readfirstlane64(fp), A.K - RB);        // wave-uniform; the DMA offsets are relative to it
    const bool dma = __all((A.K >= RB) && (fp >= readfirstlane64(fp)) && (fp + (long long)m - base < (1ll << 24)));
    if (dma) {
        const char *sbase = reinterpret_cast<const char *>(A.stream + base * 7);
        const int dw = min(lane / PPW, WPI - 1);                      // lanes 56..63 re-fetch a piece of the second window
        const int dp = (lane < PPW * WPI) ? lane - dw * PPW : 0;
        const unsigned img_base = (unsigned)(size_t)((__attribute__((address_space(3))) char *)img);
        const char *mine = img + (lane / WPI) * IMG + (lane % WPI) * (PPW * 16);   // this lane's window in the image
        for (int r0 = 0; r0 <= rmax; r0 += RB) {
            // the trip's RB knots of window i start at knot fp + min(r0, m) (rows past m repeat knot m: the tail row is knot
            // m's reading under the update time), pulled back so that the piece stays inside the stream
            const long long st = min(fp + (long long)min(r0, m), A.K - RB);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the previous trip's image reads have retired
            srel[lane] = (int)(st - base);
            wave_lds_fence();
            unsigned so[NI];                                          // all offsets first: one LDS round trip, then 32 DMA issues
#pragma unroll
            for (int j = 0; j < NI; ++j) so[j] = (unsigned)srel[j * WPI + dw];
#pragma unroll
            for (int j = 0; j < NI; ++j) glds16(so[j] * 56u + (unsigned)dp * 16u, sbase, img_base + j * IMG);
            wait_vmcnt<0>();
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < RB; i++) {
                const int r = r0 + i;
                const int jj = (int)(fp + (long long)min(r, m) - st);             // 0 .. RB - 1 for every row that is stored
                const double *src = reinterpret_cast<const double *>(mine + min(max(jj, 0), RB - 1) * 56);
                double v[7];
#pragma unroll
                for (int k = 0; k < 7; k++) v[k] = src[k];
                if (r == 0) v[0] = start_t;
                if (tail && r == m + 1) v[0] = T;
                if (r <= rows) {
                    // non-temporal row stores: 1.17-1.21 -> 1.13 ms per 1 M x 50 (the same hint on the LDS-DMA reads, whose pieces
                    // share lines with the next trip: 1.20 -> 1.35 ms)
#pragma unroll
                    for (int k = 0; k < 7; k++) __builtin_nontemporal_store(v[k], tb + (long long)r * A.ss + k * 64);
                }
            }
        }
        return;
    }
    // per-lane loads (round 3): RB rows per trip, a lane consumes RB x 56 contiguous bytes of the stream while its 128-byte
    // lines are in flight / fresh in the L1
    for (int r0 = 0; r0 <= rmax; r0 += RB) {
        double v[RB][7];
#pragma unroll
        for (int i = 0; i < RB; i++) {
            // row r of this window: r == 0 the front reading under the window's start stamp; 1 .. m stream knots; m + 1 the tail
            const int rr = min(r0 + i, rows);
            const double *kp = A.stream + (fp + min(rr, m)) * 7;
#pragma unroll
            for (int k = 0; k < 7; k++) v[i][k] = kp[k];
            if (rr == 0) v[i][0] = start_t;
            if (tail && rr == m + 1) v[i][0] = T;
        }
#pragma unroll
        for (int i = 0; i < RB; i++) {
            const int r = r0 + i;
            if (r <= rows) {
#pragma unroll
                for (int k = 0; k < 7; k++) tb[(long long)r * A.ss + k * 64] = v[i][k];
            }
        }
    }
}


}  // namespace
