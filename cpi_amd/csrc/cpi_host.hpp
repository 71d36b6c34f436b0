// cpi_host.hpp -- C++ host facade above the C-ABI (include/cpi_amd.h), shaped like the reference's
// classes so caller code written like GraphSolver_IMU.cpp:43-75 / 97-130 keeps its structure:
//
//   cpi_host::CpiV1 cpi(sigma_g, sigma_wg, sigma_a, sigma_wa);          // CpiV1.h:55
//   cpi.setLinearizationPoints(bg_K, ba_K, q_K, gravity);               // CpiBase.h:73
//   while (...) cpi.feed_IMU(t0, t1, w0, a0, w1, a1);                   // CpiBase.h:86
//   use cpi.DT, cpi.alpha_tau, cpi.beta_tau, cpi.q_k2tau, cpi.J_q ... cpi.P_meas   // CpiBase.h:99-124
//
// Differences forced by the device boundary: feed_IMU() only records the interval; the recursion runs on the GPU when a
// result member is READ (round 5: the members are lazy -- code shaped like GraphSolver_IMU.cpp:43-75 compiles unchanged, no
// finalize() between the feed_IMU loop and `ImuFactorCPIv1(..., cpi.P_meas, cpi.DT, cpi.grav, cpi.alpha_tau, ...)`), when
// finalize(ctx) is called, or through CpiBatch, which flushes many windows in one launch -- the intended use for more than a
// handful of windows.  Matrices are plain column-major arrays (Eigen::Map them if needed).
// Header-only; link with -lcpi_amd.  No CPU fallback: errors throw std::runtime_error.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cpi_amd.h"

namespace cpi_host {

typedef std::array<double, 3> Vec3;
typedef std::array<double, 4> Vec4;
typedef std::array<double, 9> Mat3;      // column-major
typedef std::array<double, 225> Mat15;   // column-major

class Context {
public:
    explicit Context(int device = -1, void *stream = nullptr) {
        if (cpi_ctx_create(device, stream, &ctx_) != CPI_OK) throw std::runtime_error(cpi_last_error(nullptr));
    }
    ~Context() { cpi_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    cpi_ctx *get() const { return ctx_; }
    void check(int rc) const { if (rc != CPI_OK) throw std::runtime_error(cpi_last_error(ctx_)); }
private:
    cpi_ctx *ctx_ = nullptr;
};

// A set of devices of this process (include/cpi_amd.h "Device sets"): one context + stream + RCCL rank per GPU.
// Windows shard as contiguous blocks (bounds(W, r)); every rank runs the ordinary entries on ctx(r) with pointers into
// its own device's memory; gather() is the one exchange step (each peer sends its output slab straight to the root).
class DeviceGroup {
public:
    explicit DeviceGroup(int n, const int *devices = nullptr) {
        if (cpi_group_create(n, devices, &g_) != CPI_OK) throw std::runtime_error(cpi_group_last_error(nullptr));
    }
    explicit DeviceGroup(cpi_group *adopt) : g_(adopt) { if (!g_) throw std::runtime_error("DeviceGroup: null group"); }   // takes ownership
    ~DeviceGroup() { cpi_group_destroy(g_); }
    DeviceGroup(const DeviceGroup &) = delete;
    DeviceGroup &operator=(const DeviceGroup &) = delete;
    DeviceGroup(DeviceGroup &&o) noexcept : g_(o.g_) { o.g_ = nullptr; }
    int size() const { return cpi_group_size(g_); }
    cpi_ctx *ctx(int rank) const { return cpi_group_ctx(g_, rank); }
    void bounds(int64_t W, int rank, int64_t &lo, int64_t &hi) const { cpi_shard_bounds(W, rank, size(), &lo, &hi); }
    void check(int rc) const { if (rc != CPI_OK) throw std::runtime_error(cpi_group_last_error(g_)); }
    void gather(int root, int64_t W, const cpi_outputs *local, const cpi_outputs &root_out) { check(cpi_group_gather(g_, root, W, local, &root_out)); }
    // The exchange inside ONE batch (include/cpi_amd.h: cpi_group_gather_chunk): every block in `chunks` sub-blocks
    // (chunk_bounds), sub-block c on the wire -- on the group's exchange streams -- while the kernels of sub-block c + 1 run.
    //   for (c = 0; c < chunks; c++) { for every rank r: enqueue the entries for chunk_bounds(W, r, c, chunks) on ctx(r);
    //                                  gather_chunk(root, W, c, chunks, local_c, root_out); }      // the last call joins
    void chunk_bounds(int64_t W, int rank, int chunk, int chunks, int64_t &lo, int64_t &hi) const { cpi_shard_chunk_bounds(W, rank, size(), chunk, chunks, &lo, &hi); }
    void gather_chunk(int root, int64_t W, int chunk, int chunks, const cpi_outputs *local_chunk, const cpi_outputs &root_out) {
        check(cpi_group_gather_chunk(g_, root, W, chunk, chunks, local_chunk, &root_out));
    }
    void synchronize() { check(cpi_group_synchronize(g_)); }
    int last_gather_messages() const { return cpi_group_last_gather_messages(g_); }   // per peer; 1 = slab path
private:
    cpi_group *g_ = nullptr;
};

// Results of one window: the public members of CpiBase / CpiV2, as plain values.
struct CpiResult {
    double DT = 0;
    Vec3 alpha_tau{}, beta_tau{};
    Vec4 q_k2tau{{0, 0, 0, 1}};
    Mat3 J_q{}, J_a{}, J_b{}, H_a{}, H_b{}, O_a{}, O_b{};
    Mat15 P_meas{};
};

// The context a lazy read runs on when the preintegrator was not bound to one (CpiBase::bind): one per THREAD, on the device
// that is current when the thread first reads a result.  Calls on one context are not re-entrant, and the reference's result
// members are plain per-object data that different threads may read side by side -- so reads of DIFFERENT preintegrators from
// different threads each use their own thread's context (stream, staging) and do not race.  ONE preintegrator read from two
// threads at once is still a data race on its members the first time (the read runs the recursion and stores the results):
// read it once before sharing it, or guard it -- INTEGRATION.md section 3.
inline const Context &default_context() {
    static thread_local Context ctx;
    return ctx;
}

class CpiBase;
// A result member of CpiBase (CpiBase.h:99-124: DT, alpha_tau, ... P_meas).  The reference's members are live after every
// feed_IMU; here the recorded intervals run on the GPU the first time ANY member is read after a feed_IMU (one launch for the
// whole window so far), so caller code keeps the reference's shape.  Reading = conversion to const T &, get(), and for the
// array members operator[], begin() / end(), data().  Assignment stores a value and runs nothing.
template <class T>
class Lazy {
public:
    explicit Lazy(const CpiBase *owner, const T &init = T{}, bool mean = false) : v_(init), owner_(owner), mean_(mean) {}
    Lazy(const Lazy &) = delete;                       // members of ONE preintegrator: CpiBase's copy operations re-bind them
    Lazy &operator=(const Lazy &) = delete;
    Lazy &operator=(const T &v) { v_ = v; return *this; }
    operator const T &() const { sync(); return v_; }
    const T &get() const { sync(); return v_; }
    template <class U = T> auto operator[](size_t i) const -> decltype(std::declval<const U &>()[i]) { sync(); return v_[i]; }
    template <class U = T> auto begin() const -> decltype(std::declval<const U &>().begin()) { sync(); return v_.begin(); }
    template <class U = T> auto end() const -> decltype(std::declval<const U &>().end()) { sync(); return v_.end(); }
    template <class U = T> auto data() const -> decltype(std::declval<const U &>().data()) { sync(); return v_.data(); }
    template <class U = T> auto size() const -> decltype(std::declval<const U &>().size()) { return v_.size(); }
private:
    friend class CpiBase;
    inline void sync() const;
    T v_;
    const CpiBase *owner_;
    bool mean_;            // one of DT / alpha_tau / beta_tau / q_k2tau: valid after a mean-only flush
};

class CpiBase {
public:
    CpiBase(int model, double sigma_w, double sigma_wb, double sigma_a, double sigma_ab, bool imu_avg_ = false)
        : imu_avg(imu_avg_), model_(model) { sig_[0] = sigma_w; sig_[1] = sigma_wb; sig_[2] = sigma_a; sig_[3] = sigma_ab; }
    CpiBase(const CpiBase &o) { *this = o; }
    CpiBase &operator=(const CpiBase &o) {
        if (this == &o) return *this;
        imu_avg = o.imu_avg; state_transition_jacobians = o.state_transition_jacobians;
        b_w_lin = o.b_w_lin; b_a_lin = o.b_a_lin; q_k_lin = o.q_k_lin; grav = o.grav;
        knots_ = o.knots_; model_ = o.model_; ctx_ = o.ctx_; dirty_ = o.dirty_; means_only_ = o.means_only_;
        for (int i = 0; i < 4; i++) sig_[i] = o.sig_[i];
        put(o.peek());
        return *this;
    }
    virtual ~CpiBase() {}
    // the context lazy reads run on (default: default_context()); the Context must outlive the reads
    void bind(const Context &ctx) { ctx_ = &ctx; }
    // Like the reference (CpiBase.h:73-80, which only stores the values; the recursion reads them at every feed_IMU), the
    // linearisation point belongs BEFORE the first feed_IMU.  Here the recursion runs at the first read, with the values current
    // then; a call after results were read marks them stale, so the next read recomputes the window at the new point (the
    // reference would keep integrating the old prefix with the old point -- a use it never makes).  Assigning the public members
    // b_w_lin / b_a_lin / q_k_lin / grav / imu_avg directly is NOT tracked: call invalidate() afterwards.
    void setLinearizationPoints(const Vec3 &b_w_lin_, const Vec3 &b_a_lin_, const Vec4 &q_k_lin_ = Vec4{{0, 0, 0, 0}},
                                const Vec3 &grav_ = Vec3{{0, 0, 0}}) {
        b_w_lin = b_w_lin_; b_a_lin = b_a_lin_; q_k_lin = q_k_lin_; grav = grav_;
        if (!knots_.empty()) dirty_ = true;
    }
    void invalidate() { if (!knots_.empty()) dirty_ = true; }
    // Records interval [t_0, t_1] with readings (w_m_0, a_m_0) at t_0 and (w_m_1, a_m_1) at t_1.
    // Differences from CpiV1::feed_IMU / CpiV2::feed_IMU (CpiV1.h:62-74):
    //  * t_1 - t_0 < 0: the reference's feed_IMU integrates the interval with the negative dt (only its caller,
    //    GraphSolver_IMU.cpp:52, skips it); here -- as in the C-ABI -- such an interval is SKIPPED, like dt == 0.
    //  * imu_avg == false: the closing reading (w_m_1, a_m_1) takes no part in the arithmetic (CpiV1.h:77-86), and callers
    //    written like the reference's non-averaging use leave it at its default.  The interval then chains to the previous
    //    one whenever the TIMES chain (t_0 == previous t_1): the previous closing knot is overwritten with (w_m_0, a_m_0)
    //    and every interval costs ONE knot (no NaN separator, no 3 knots per interval -- the N <= 65535 limit of a window
    //    would otherwise be reached after ~21 k calls).  Set imu_avg before the first feed_IMU.
    void feed_IMU(double t_0, double t_1, const Vec3 &w_m_0, const Vec3 &a_m_0, const Vec3 &w_m_1 = Vec3{{0, 0, 0}},
                  const Vec3 &a_m_1 = Vec3{{0, 0, 0}}) {
        const size_t n = knots_.size() / 7;
        bool chained = false;
        if (n) {
            double *k = &knots_[(n - 1) * 7];
            if (!imu_avg && k[0] == t_0) {   // times chain, closing readings unused: this interval's reading opens it
                for (int i = 0; i < 3; i++) { k[1 + i] = w_m_0[i]; k[4 + i] = a_m_0[i]; }
                chained = true;
                touch();
            } else {
                chained = k[0] == t_0 && k[1] == w_m_0[0] && k[2] == w_m_0[1] && k[3] == w_m_0[2] && k[4] == a_m_0[0] &&
                          k[5] == a_m_0[1] && k[6] == a_m_0[2];
            }
        }
        if (!chained) {
            // The reference's feed_IMU only ever uses t_1 - t_0, so intervals need not chain.  A knot whose
            // time is NaN is a separator: both intervals touching it have a NaN dt and are skipped by the kernels.
            if (n) push(std::numeric_limits<double>::quiet_NaN(), Vec3{{0, 0, 0}}, Vec3{{0, 0, 0}});
            push(t_0, w_m_0, a_m_0);
        }
        push(t_1, w_m_1, a_m_1);
    }
    // Runs this single window on the GPU and fills the result members (what a first read of any member does by itself).
    void finalize(const Context &ctx) {
        cpi_params p = params();
        const double lin[6] = { b_w_lin[0], b_w_lin[1], b_w_lin[2], b_a_lin[0], b_a_lin[1], b_a_lin[2] };
        CpiResult r;
        cpi_outputs o = outputs_of(r);
        const int32_t n = knots_.empty() ? 0 : (int32_t)(knots_.size() / 7 - 1);
        static const double zero_knot[7] = { 0, 0, 0, 0, 0, 0, 0 };
        ctx.check(cpi_preintegrate_batch_host(ctx.get(), &p, 1, n, knots_.empty() ? zero_knot : knots_.data(), nullptr, nullptr,
                                              n + 1, lin, q_k_lin.data(), &o));
        set_result(r);
    }
    // the result members as plain values (runs the pending intervals first) / stored from outside (CpiBatch)
    CpiResult result() const { ensure(true); return peek(); }
    void set_result(const CpiResult &r) { put(r); dirty_ = false; means_only_ = false; }
    // CpiBatch::flush_means: only DT / alpha_tau / beta_tau / q_k2tau were computed.  The Jacobian and covariance members are
    // NOT valid for the recorded intervals: the first read of one of them runs the whole window (finalize), reads of the four
    // means do not.
    void set_means(double DT_, const Vec3 &alpha, const Vec3 &beta, const Vec4 &q) {
        DT.v_ = DT_; alpha_tau.v_ = alpha; beta_tau.v_ = beta; q_k2tau.v_ = q; dirty_ = false; means_only_ = true;
    }
    cpi_params params() const {
        cpi_params p{};
        p.sigma_w = sig_[0]; p.sigma_wb = sig_[1]; p.sigma_a = sig_[2]; p.sigma_ab = sig_[3];
        for (int i = 0; i < 3; i++) p.grav[i] = grav[i];
        p.model = model_; p.imu_avg = imu_avg ? 1 : 0;
        p.state_transition_jacobians = state_transition_jacobians ? 1 : 0;
        p.lanes_per_window = 0;
        return p;
    }
    static cpi_outputs outputs_of(CpiResult &r) {
        cpi_outputs o{};
        o.DT = &r.DT; o.alpha = r.alpha_tau.data(); o.beta = r.beta_tau.data(); o.q = r.q_k2tau.data();
        o.J_q = r.J_q.data(); o.J_a = r.J_a.data(); o.J_b = r.J_b.data(); o.H_a = r.H_a.data(); o.H_b = r.H_b.data();
        o.O_a = r.O_a.data(); o.O_b = r.O_b.data(); o.P = r.P_meas.data();
        return o;
    }
    const std::vector<double> &knots() const { return knots_; }
    int model() const { return model_; }
    // the model an ImuFactorCPI built from this measurement evaluates: the Forster comparator's result is wrapped in
    // an ImuFactorCPIv1 by the reference (GraphSolver_IMU.cpp:227-231)
    int factor_model() const { return model_ == CPI_MODEL_FORSTER ? CPI_MODEL_V1 : model_; }

    bool imu_avg = false;
    bool state_transition_jacobians = true;  // CpiV2.h:58
    Vec3 b_w_lin{}, b_a_lin{};
    Vec4 q_k_lin{};
    Vec3 grav{};
    // CpiBase.h:99-124 (+ CpiV2.h O_a / O_b): computed on first read after a feed_IMU
    Lazy<double> DT{this, 0.0, true};
    Lazy<Vec3> alpha_tau{this, Vec3{}, true}, beta_tau{this, Vec3{}, true};
    Lazy<Vec4> q_k2tau{this, Vec4{{0, 0, 0, 1}}, true};
    Lazy<Mat3> J_q{this}, J_a{this}, J_b{this}, H_a{this}, H_b{this}, O_a{this}, O_b{this};
    Lazy<Mat15> P_meas{this};

protected:
    void push(double t, const Vec3 &w, const Vec3 &a) {
        knots_.push_back(t);
        for (int i = 0; i < 3; i++) knots_.push_back(w[i]);
        for (int i = 0; i < 3; i++) knots_.push_back(a[i]);
        dirty_ = true;
    }
    void touch() { dirty_ = true; }              // a recorded knot was rewritten in place
    std::vector<double> knots_;
private:
    template <class T> friend class Lazy;
    // full: the member being read is a Jacobian / covariance (not one of the four means)
    void ensure(bool full = true) const { if (dirty_ || (full && means_only_)) const_cast<CpiBase *>(this)->finalize(ctx_ ? *ctx_ : default_context()); }
    CpiResult peek() const {                     // the stored values, without running anything
        CpiResult r;
        r.DT = DT.v_; r.alpha_tau = alpha_tau.v_; r.beta_tau = beta_tau.v_; r.q_k2tau = q_k2tau.v_;
        r.J_q = J_q.v_; r.J_a = J_a.v_; r.J_b = J_b.v_; r.H_a = H_a.v_; r.H_b = H_b.v_; r.O_a = O_a.v_; r.O_b = O_b.v_;
        r.P_meas = P_meas.v_;
        return r;
    }
    void put(const CpiResult &r) {
        DT.v_ = r.DT; alpha_tau.v_ = r.alpha_tau; beta_tau.v_ = r.beta_tau; q_k2tau.v_ = r.q_k2tau;
        J_q.v_ = r.J_q; J_a.v_ = r.J_a; J_b.v_ = r.J_b; H_a.v_ = r.H_a; H_b.v_ = r.H_b; O_a.v_ = r.O_a; O_b.v_ = r.O_b;
        P_meas.v_ = r.P_meas;
    }
    int model_ = CPI_MODEL_V1;
    double sig_[4] = {0, 0, 0, 0};
    const Context *ctx_ = nullptr;
    bool dirty_ = false;                         // intervals recorded since the result members were last computed
    bool means_only_ = false;                    // the last computation (CpiBatch::flush_means) filled the four means only
};
template <class T> inline void Lazy<T>::sync() const { owner_->ensure(!mean_); }

class CpiV1 : public CpiBase {
public:
    CpiV1(double sigma_w, double sigma_wb, double sigma_a, double sigma_ab, bool imu_avg_ = false)
        : CpiBase(CPI_MODEL_V1, sigma_w, sigma_wb, sigma_a, sigma_ab, imu_avg_) {}
};
class CpiV2 : public CpiBase {
public:
    CpiV2(double sigma_w, double sigma_wb, double sigma_a, double sigma_ab, bool imu_avg_ = false)
        : CpiBase(CPI_MODEL_V2, sigma_w, sigma_wb, sigma_a, sigma_ab, imu_avg_) {}
};

// The "Forster discrete" comparator: what GraphSolver::createimufactor_discrete (GraphSolver_IMU.cpp:141-232) builds
// with gtsam::PreintegratedCombinedMeasurements -- the four covariances of :152-155 are the constructor's sigmas^2, the
// bias estimate of :162 goes to setLinearizationPoints(bg_K, ba_K), the loop calls integrateMeasurement(acc, omega, dt)
// (:180, :194; GTSAM's argument order).  After finalize() / CpiBatch::flush() the result members hold what :204-225
// derive from the GTSAM object (alpha_tau = deltaPij, beta_tau = deltaVij, q_k2tau = rot_2_quat(deltaRij^T),
// J_q = -delRdelBiasOmega, J_b / J_a = delV / delP delBiasOmega, H_b / H_a = delV / delP delBiasAcc, P_meas = the
// block-swapped preintMeasCov), ready for ImuFactorCPI (:227-231).  GTSAM is not part of the reference tree: parity
// of this model is unpinned (oracle/forster_oracle.c).
class ForsterDiscrete : public CpiBase {
public:
    ForsterDiscrete(double sigma_g, double sigma_wg, double sigma_a, double sigma_wa)
        : CpiBase(CPI_MODEL_FORSTER, sigma_g, sigma_wg, sigma_a, sigma_wa, false) {}
    // reading (measuredAcc, measuredOmega) held over the next dt seconds; dt <= 0 is ignored (GraphSolver_IMU.cpp:171)
    void integrateMeasurement(const Vec3 &measuredAcc, const Vec3 &measuredOmega, double dt) {
        if (!(dt > 0)) return;
        if (knots_.empty()) push(0.0, measuredOmega, measuredAcc);
        else {   // the closing knot of the previous interval becomes the opening knot of this one: give it this reading
            double *k = &knots_[knots_.size() - 7];
            for (int i = 0; i < 3; i++) { k[1 + i] = measuredOmega[i]; k[4 + i] = measuredAcc[i]; }
            touch();
        }
        // intervals are stored as knot times; t += dt reproduces dt to within one rounding of the running time
        t_ += dt;
        push(t_, measuredOmega, measuredAcc);
    }
    double deltaTij() const { return DT; }
private:
    double t_ = 0.0;
};

// Collects many recorded windows (same model / flags / gravity) and runs them in ONE launch.
class CpiBatch {
public:
    void add(CpiBase *w) { win_.push_back(w); }
    void flush(const Context &ctx) {
        if (win_.empty()) return;
        const int64_t W = (int64_t)win_.size();
        std::vector<double> knots, lin(W * 6), qk(W * 4);
        std::vector<int64_t> first(W);
        std::vector<int32_t> count(W);
        int32_t N = 0;
        for (int64_t w = 0; w < W; w++) {
            const std::vector<double> &k = win_[w]->knots();
            first[w] = (int64_t)(knots.size() / 7);
            count[w] = k.empty() ? 0 : (int32_t)(k.size() / 7 - 1);
            if (count[w] > N) N = count[w];
            if (k.empty()) knots.insert(knots.end(), 7, 0.0); else knots.insert(knots.end(), k.begin(), k.end());
            for (int i = 0; i < 3; i++) { lin[w * 6 + i] = win_[w]->b_w_lin[i]; lin[w * 6 + 3 + i] = win_[w]->b_a_lin[i]; }
            for (int i = 0; i < 4; i++) qk[w * 4 + i] = win_[w]->q_k_lin[i];
        }
        cpi_params p = win_[0]->params();
        std::vector<double> DT(W), al(W * 3), be(W * 3), q(W * 4), Jq(W * 9), Ja(W * 9), Jb(W * 9), Ha(W * 9), Hb(W * 9),
            Oa(W * 9), Ob(W * 9), P(W * 225);
        cpi_outputs o{ DT.data(), al.data(), be.data(), q.data(), Jq.data(), Ja.data(), Jb.data(), Ha.data(), Hb.data(),
                       Oa.data(), Ob.data(), P.data() };
        // windows of equal length lie back to back = the dense layout: that entry pipelines upload / kernels / download
        bool dense = true;
        for (int64_t w = 0; w < W && dense; w++) dense = (count[w] == N) && !win_[w]->knots().empty();
        ctx.check(cpi_preintegrate_batch_host(ctx.get(), &p, W, N, knots.data(), dense ? nullptr : first.data(), dense ? nullptr : count.data(),
                                              (int64_t)(knots.size() / 7), lin.data(), qk.data(), &o));
        for (int64_t w = 0; w < W; w++) {
            CpiResult r;
            r.DT = DT[w];
            for (int i = 0; i < 3; i++) { r.alpha_tau[i] = al[w * 3 + i]; r.beta_tau[i] = be[w * 3 + i]; }
            for (int i = 0; i < 4; i++) r.q_k2tau[i] = q[w * 4 + i];
            for (int i = 0; i < 9; i++) {
                r.J_q[i] = Jq[w * 9 + i]; r.J_a[i] = Ja[w * 9 + i]; r.J_b[i] = Jb[w * 9 + i]; r.H_a[i] = Ha[w * 9 + i];
                r.H_b[i] = Hb[w * 9 + i]; r.O_a[i] = Oa[w * 9 + i]; r.O_b[i] = Ob[w * 9 + i];
            }
            for (int i = 0; i < 225; i++) r.P_meas[i] = P[w * 225 + i];
            win_[w]->set_result(r);
        }
        win_.clear();
    }
    // Mean outputs only (DT, alpha_tau, beta_tau, q_k2tau): the HBM-bound request.  The recorded windows are written
    // straight into the TILED layout (include/cpi_amd.h: tiles[ceil(W/64)][N+1][7][64], knot s of window w at
    // (((w / 64) (N+1) + s) 7 + k) 64 + w % 64 -- no dense copy is ever made) with their own interval counts, and go
    // through cpi_preintegrate_tiled_batch_host (chunked upload / kernel / download pipeline).  Models 1 and 2.
    void flush_means(const Context &ctx) {
        if (win_.empty()) return;
        const int64_t W = (int64_t)win_.size();
        std::vector<int32_t> count(W);
        int32_t N = 0;
        for (int64_t w = 0; w < W; w++) {
            const std::vector<double> &k = win_[w]->knots();
            count[w] = k.empty() ? 0 : (int32_t)(k.size() / 7 - 1);
            if (count[w] > N) N = count[w];
        }
        const int64_t nb = (W + 63) / 64;
        std::vector<double> tiles((size_t)nb * (N + 1) * 448, 0.0), lin(W * 6), qk(W * 4);
        for (int64_t w = 0; w < W; w++) {
            const std::vector<double> &k = win_[w]->knots();
            double *col = &tiles[(size_t)(w / 64) * (N + 1) * 448 + (size_t)(w % 64)];
            for (int32_t s = 0; s <= count[w] && !k.empty(); s++)
                for (int f = 0; f < 7; f++) col[((size_t)s * 7 + f) * 64] = k[(size_t)s * 7 + f];
            for (int i = 0; i < 3; i++) { lin[w * 6 + i] = win_[w]->b_w_lin[i]; lin[w * 6 + 3 + i] = win_[w]->b_a_lin[i]; }
            for (int i = 0; i < 4; i++) qk[w * 4 + i] = win_[w]->q_k_lin[i];
        }
        cpi_params p = win_[0]->params();
        std::vector<double> DT(W), al(W * 3), be(W * 3), q(W * 4);
        cpi_outputs o{};
        o.DT = DT.data(); o.alpha = al.data(); o.beta = be.data(); o.q = q.data();
        ctx.check(cpi_preintegrate_tiled_batch_host(ctx.get(), &p, W, N, tiles.data(), count.data(), lin.data(), qk.data(), &o));
        for (int64_t w = 0; w < W; w++)
            win_[w]->set_means(DT[w], Vec3{{al[w * 3], al[w * 3 + 1], al[w * 3 + 2]}}, Vec3{{be[w * 3], be[w * 3 + 1], be[w * 3 + 2]}},
                               Vec4{{q[w * 4], q[w * 4 + 1], q[w * 4 + 2], q[w * 4 + 3]}});
        win_.clear();
    }
private:
    std::vector<CpiBase *> win_;
};

// ---- caller-side data formats --------------------------------------------
// The reference's simulated-IMU wire format, one reading per line "wx wy wz ax ay az 0 t_ms"
// (cpi_compare/src/sim/SimParser.h:130-193: fields split on single spaces, empty fields skipped, column 8 is
// the stamp in milliseconds).  Returns knot records {t[s], w[3], a[3]}.
inline std::vector<double> parse_imu_text(const std::string &text) {
    std::vector<double> knots;
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        double f[8];
        int nf = 0;
        size_t p = pos;
        while (p < eol && nf < 8) {
            while (p < eol && text[p] == ' ') p++;
            if (p >= eol) break;
            size_t q = p;
            while (q < eol && text[q] != ' ') q++;
            f[nf++] = std::atof(text.substr(p, q - p).c_str());
            p = q;
        }
        if (nf == 8) {
            knots.push_back(1e-3 * f[7]);
            for (int i = 0; i < 6; i++) knots.push_back(f[i]);
        }
        pos = eol + 1;
    }
    return knots;
}

// Cutting ONE IMU stream into preintegration windows at successive update times exactly like
// GraphSolver::createimufactor_cpi_v1/v2 (GraphSolver_IMU.cpp:50-69): whole intervals while
// imu_times[1] <= updatetime, then the partial tail interval with the front reading repeated, after which the
// front stamp is overwritten by the update time.  Output is the knots/first/count layout of cpi_amd.h.
struct WindowSet {
    std::vector<double> knots;
    std::vector<int64_t> first;
    std::vector<int32_t> count;
    int32_t max_count = 0;
};
inline WindowSet assemble_windows(const std::vector<double> &stream, const std::vector<double> &update_times) {
    WindowSet ws;
    const size_t K = stream.size() / 7;
    if (K == 0) return ws;
    size_t front = 0;
    double front_t = stream[0];
    auto push = [&](double t, const double *r) { ws.knots.push_back(t); for (int i = 1; i < 7; i++) ws.knots.push_back(r[i]); };
    for (double T : update_times) {
        ws.first.push_back((int64_t)(ws.knots.size() / 7));
        push(front_t, &stream[front * 7]);
        int32_t n = 0;
        while (K - front > 1 && stream[(front + 1) * 7] <= T) {
            front++;
            front_t = stream[front * 7];
            push(front_t, &stream[front * 7]);   // dt < 0 intervals stay in the list: the kernels skip them
            n++;
        }
        if (T - front_t > 0) { push(T, &stream[front * 7]); front_t = T; n++; }
        ws.count.push_back(n);
        if (n > ws.max_count) ws.max_count = n;
    }
    return ws;
}

// The same cutting written STRAIGHT into the tiled layout of cpi_preintegrate_tiled_batch (mean-only requests; DESIGN.md
// 3.1a): pass 1 walks the deque loop and records where each window starts and how long it is, pass 2 places knot s of
// window w at its tile slot.  No intermediate knots / first / count copy.  N = the largest count (or min_N if larger).
struct TiledWindowSet {
    std::vector<double> tiles;     // [ceil(W/64)][N+1][7][64]
    std::vector<int32_t> count;    // [W]
    int64_t W = 0;
    int32_t N = 0;
};
inline TiledWindowSet assemble_windows_tiled(const std::vector<double> &stream, const std::vector<double> &update_times, int32_t min_N = 0) {
    TiledWindowSet ts;
    const size_t K = stream.size() / 7;
    if (K == 0) return ts;
    struct Win { size_t front0; double start_t; int32_t whole; bool tail; double T; };
    std::vector<Win> win;
    win.reserve(update_times.size());
    size_t front = 0;
    double front_t = stream[0];
    int32_t N = min_N;
    for (double T : update_times) {
        Win w{front, front_t, 0, false, T};
        while (K - front > 1 && stream[(front + 1) * 7] <= T) { front++; front_t = stream[front * 7]; w.whole++; }
        if (T - front_t > 0) { w.tail = true; front_t = T; }
        win.push_back(w);
        const int32_t n = w.whole + (w.tail ? 1 : 0);
        ts.count.push_back(n);
        if (n > N) N = n;
    }
    ts.W = (int64_t)win.size();
    ts.N = N;
    const size_t nb = (size_t)((ts.W + 63) / 64), tstride = (size_t)(N + 1) * 448;
    ts.tiles.assign(nb * tstride, 0.0);
    for (size_t w = 0; w < win.size(); w++) {
        double *col = &ts.tiles[(w / 64) * tstride + (w % 64)];
        auto put = [&](int32_t s, double t, const double *r) { col[((size_t)s * 7) * 64] = t; for (int f = 1; f < 7; f++) col[((size_t)s * 7 + f) * 64] = r[f]; };
        put(0, win[w].start_t, &stream[win[w].front0 * 7]);
        for (int32_t s = 1; s <= win[w].whole; s++) put(s, stream[(win[w].front0 + s) * 7], &stream[(win[w].front0 + s) * 7]);
        if (win[w].tail) put(win[w].whole + 1, win[w].T, &stream[(win[w].front0 + win[w].whole) * 7]);
    }
    return ts;
}

// ---- the caller's loop for MANY windows at once --------------------------------------------------------------------
// What GraphSolver keeps between two states is a deque of IMU readings (GraphSolver.h: imu_times / imu_linaccs /
// imu_angvel, filled by addmeasurement_imu); createimufactor_cpi_v1 / _v2 (GraphSolver_IMU.cpp:34-134) walk it up to the
// update time, feed a CpiV1 / CpiV2 and read its members.  ImuStream is that deque as one array, and preintegrate() is that
// loop for EVERY update time in one call: the stream goes to the device once, the kernels cut the windows themselves
// (cpi_preintegrate_stream_host), one CpiResult per update time comes back -- window u covers (update[u-1], update[u]]
// (the first one starts at the stream's first reading), its tail interval holds the front reading until the update time.
class ImuStream {
public:
    void push(double t, const Vec3 &w, const Vec3 &a) {           // GraphSolver::addmeasurement_imu
        knots_.push_back(t);
        for (int i = 0; i < 3; i++) knots_.push_back(w[i]);
        for (int i = 0; i < 3; i++) knots_.push_back(a[i]);
    }
    void assign(const std::vector<double> &knots) { knots_ = knots; }   // e.g. parse_imu_text's records
    size_t size() const { return knots_.size() / 7; }
    const std::vector<double> &knots() const { return knots_; }
    // prm: cpi_params as CpiBase::params() builds them (model, sigmas, gravity, flags).  lin [U][6] = b_w_lin, b_a_lin per window;
    // q_k_lin [U][4] (model 2; may be empty otherwise).  counts (optional) receives every window's interval count.
    // A window longer than max_intervals (default: as long as the stream) is an error, not a truncation.
    std::vector<CpiResult> preintegrate(const Context &ctx, const cpi_params &prm, const std::vector<double> &update_times,
                                        const std::vector<double> &lin, const std::vector<double> &q_k_lin = std::vector<double>(),
                                        std::vector<int32_t> *counts = nullptr, int32_t max_intervals = 0) const {
        const int64_t U = (int64_t)update_times.size(), K = (int64_t)size();
        if ((int64_t)lin.size() != U * 6) throw std::runtime_error("ImuStream::preintegrate: lin must hold 6 doubles per update time");
        if (!q_k_lin.empty() && (int64_t)q_k_lin.size() != U * 4) throw std::runtime_error("ImuStream::preintegrate: q_k_lin must hold 4 doubles per update time");
        std::vector<CpiResult> res((size_t)U);
        if (U == 0) return res;
        const int32_t N = max_intervals > 0 ? max_intervals : (int32_t)std::min<int64_t>(std::max<int64_t>(K, 1), 65535);
        std::vector<double> DT(U), al(U * 3), be(U * 3), q(U * 4), Jq(U * 9), Ja(U * 9), Jb(U * 9), Ha(U * 9), Hb(U * 9), Oa(U * 9),
            Ob(U * 9), P(U * 225);
        cpi_outputs o{ DT.data(), al.data(), be.data(), q.data(), Jq.data(), Ja.data(), Jb.data(), Ha.data(), Hb.data(),
                       prm.model == CPI_MODEL_V2 ? Oa.data() : nullptr, prm.model == CPI_MODEL_V2 ? Ob.data() : nullptr, P.data() };
        std::vector<int32_t> cnt((size_t)U);
        ctx.check(cpi_preintegrate_stream_host(ctx.get(), &prm, K, knots_.data(), U, update_times.data(), N, lin.data(),
                                               q_k_lin.empty() ? nullptr : q_k_lin.data(), &o, cnt.data()));
        for (int64_t u = 0; u < U; u++)
            if (cnt[u] > N) throw std::runtime_error("ImuStream::preintegrate: window " + std::to_string(u) + " holds " + std::to_string(cnt[u]) +
                                                     " intervals, more than max_intervals = " + std::to_string(N));
        for (int64_t w = 0; w < U; w++) {
            CpiResult &r = res[w];
            r.DT = DT[w];
            for (int i = 0; i < 3; i++) { r.alpha_tau[i] = al[w * 3 + i]; r.beta_tau[i] = be[w * 3 + i]; }
            for (int i = 0; i < 4; i++) r.q_k2tau[i] = q[w * 4 + i];
            for (int i = 0; i < 9; i++) {
                r.J_q[i] = Jq[w * 9 + i]; r.J_a[i] = Ja[w * 9 + i]; r.J_b[i] = Jb[w * 9 + i]; r.H_a[i] = Ha[w * 9 + i];
                r.H_b[i] = Hb[w * 9 + i]; r.O_a[i] = Oa[w * 9 + i]; r.O_b[i] = Ob[w * 9 + i];
            }
            for (int i = 0; i < 225; i++) r.P_meas[i] = P[w * 225 + i];
        }
        if (counts) *counts = cnt;
        return res;
    }
private:
    std::vector<double> knots_;
};

// evaluateError-shaped evaluator (ImuFactorCPIv1.h:139 / ImuFactorCPIv2.h:151).  state = 16 doubles
// [q(4) bg(3) v(3) ba(3) p(3)]; error[15]; H1/H2 column-major 15x15, may be nullptr.
class ImuFactorCPI {
public:
    // built straight from a finished preintegrator, with the field->ctor mapping of GraphSolver_IMU.cpp:74-75,129-130
    explicit ImuFactorCPI(const CpiBase &cpi) : model_(cpi.factor_model()), m_(cpi.result()), grav_(cpi.grav), qk_(cpi.q_k_lin) {
        for (int i = 0; i < 3; i++) { lin_[i] = cpi.b_w_lin[i]; lin_[3 + i] = cpi.b_a_lin[i]; }
    }
    // the reference's constructor shape (ImuFactorCPIv1.h:78-81 / ImuFactorCPIv2.h:82-85 without the two keys), so that the last
    // line of createimufactor_cpi_v1 (GraphSolver_IMU.cpp:74-75) keeps its argument list:
    //   ImuFactorCPI(cpi.P_meas, cpi.DT, cpi.grav, cpi.alpha_tau, cpi.beta_tau, cpi.q_k2tau, cpi.b_a_lin, cpi.b_w_lin,
    //                cpi.J_q, cpi.J_b, cpi.J_a, cpi.H_b, cpi.H_a)                                    model 1
    //   ImuFactorCPI(..., cpi.q_k2tau, cpi.q_k_lin, cpi.b_a_lin, ..., cpi.H_a, cpi.O_b, cpi.O_a)     model 2 (:129-130)
    ImuFactorCPI(const Mat15 &covariance, double deltatime, const Vec3 &grav, const Vec3 &alpha, const Vec3 &beta, const Vec4 &q_KtoK1,
                 const Vec3 &ba_lin, const Vec3 &bg_lin, const Mat3 &J_q, const Mat3 &J_beta, const Mat3 &J_alpha, const Mat3 &H_beta,
                 const Mat3 &H_alpha)
        : model_(CPI_MODEL_V1), grav_(grav), qk_(Vec4{{0, 0, 0, 1}}) {
        m_.P_meas = covariance; m_.DT = deltatime; m_.alpha_tau = alpha; m_.beta_tau = beta; m_.q_k2tau = q_KtoK1;
        m_.J_q = J_q; m_.J_b = J_beta; m_.J_a = J_alpha; m_.H_b = H_beta; m_.H_a = H_alpha;
        for (int i = 0; i < 3; i++) { lin_[i] = bg_lin[i]; lin_[3 + i] = ba_lin[i]; }
    }
    ImuFactorCPI(const Mat15 &covariance, double deltatime, const Vec3 &grav, const Vec3 &alpha, const Vec3 &beta, const Vec4 &q_KtoK1,
                 const Vec4 &q_K_lin, const Vec3 &ba_lin, const Vec3 &bg_lin, const Mat3 &J_q, const Mat3 &J_beta, const Mat3 &J_alpha,
                 const Mat3 &H_beta, const Mat3 &H_alpha, const Mat3 &O_beta, const Mat3 &O_alpha)
        : model_(CPI_MODEL_V2), grav_(grav), qk_(q_K_lin) {
        m_.P_meas = covariance; m_.DT = deltatime; m_.alpha_tau = alpha; m_.beta_tau = beta; m_.q_k2tau = q_KtoK1;
        m_.J_q = J_q; m_.J_b = J_beta; m_.J_a = J_alpha; m_.H_b = H_beta; m_.H_a = H_alpha; m_.O_b = O_beta; m_.O_a = O_alpha;
        for (int i = 0; i < 3; i++) { lin_[i] = bg_lin[i]; lin_[3 + i] = ba_lin[i]; }
    }
    // from a window of ImuStream::preintegrate: the measurement, the model (1 / 2) and the linearisation point it was made with
    ImuFactorCPI(int model, const CpiResult &meas, const Vec3 &grav, const Vec3 &b_w_lin, const Vec3 &b_a_lin, const Vec4 &q_k_lin = Vec4{{0, 0, 0, 1}})
        : model_(model == CPI_MODEL_FORSTER ? CPI_MODEL_V1 : model), m_(meas), grav_(grav), qk_(q_k_lin) {
        for (int i = 0; i < 3; i++) { lin_[i] = b_w_lin[i]; lin_[3 + i] = b_a_lin[i]; }
    }
    void evaluateError(const Context &ctx, const double *state_i, const double *state_j, double *error, double *H1 = nullptr,
                       double *H2 = nullptr) {
        double states[32];
        for (int i = 0; i < 16; i++) { states[i] = state_i[i]; states[16 + i] = state_j[i]; }
        cpi_outputs o = CpiBase::outputs_of(m_);
        ctx.check(cpi_factor_eval_batch_host(ctx.get(), model_, grav_.data(), 1, &o, lin_, qk_.data(), states, 2, nullptr, nullptr,
                                             error, H1, H2));
    }
private:
    int model_;
    CpiResult m_;
    Vec3 grav_;
    Vec4 qk_;
    double lin_[6];
};

}  // namespace cpi_host
