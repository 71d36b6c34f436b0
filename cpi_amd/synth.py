"""Seeded synthetic IMU windows.

Signal model mirrors the reference's simulator (cpi_simulation/SCRIPT_gazebo_to_sim.m:143-149)
and its ADIS16448 launch values (cpi_compare/launch/synthetic_test.launch:13-17):
    w_m = A_w * sin(2 pi f_w t + phi_w) + b_g + sigma_g / sqrt(dt) * n
    a_m = [0,0,9.8] + A_a * sin(2 pi f_a t + phi_a) + b_a + sigma_a / sqrt(dt) * n
5 % of windows are noise-free with tiny |w| (Taylor branch of CpiV1.h:101), 1 % carry one 5x-long
gap and 1 % one repeated timestamp (dt == 0, CpiV1.h:72).  Implemented with torch so the same code
fills host arrays (tests) and device arrays (bench, 8 M-window config) without a host round trip.
"""
import math

import torch

SIGMA_G, SIGMA_WG, SIGMA_A, SIGMA_WA = 0.005, 4e-6, 0.01, 2e-4
GRAV = (0.0, 0.0, 9.8)
BASE_SEED = 20190101


def make_windows(W, N, seed=BASE_SEED, rate=200.0, device="cpu", edge_cases=True):
    """Returns knots [W, N+1, 7] = {t, w[3], a[3]}, lin [W, 6] = {b_w_lin, b_a_lin},
    q_k_lin [W, 4] (JPL, w >= 0), all float64 on `device`."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    f64 = dict(dtype=torch.float64, device=device)

    def U(shape, lo, hi):
        return lo + (hi - lo) * torch.rand(shape, generator=g, **f64)

    def Nrm(shape):
        return torch.randn(shape, generator=g, **f64)

    dt = 1.0 / rate
    idx = torch.arange(N + 1, **f64)
    t0 = 1275.1 + U((W, 1), 0.0, 600.0)
    steps = torch.ones((W, N), **f64)
    sel = torch.rand((W,), generator=g, **f64)
    pos = torch.randint(0, N, (W,), generator=g, device=device)
    if edge_cases:
        gap = sel < 0.01
        rep = (sel >= 0.01) & (sel < 0.02)
        rows = torch.arange(W, device=device)
        steps[rows[gap], pos[gap]] = 5.0
        steps[rows[rep], pos[rep]] = 0.0
    kidx = torch.cat([torch.zeros((W, 1), **f64), torch.cumsum(steps, dim=1)], dim=1)
    t = t0 + kidx / rate                       # absolute stamps: t[i+1]-t[i] carries rounding
    del idx

    A_w = U((W, 1, 3), 0.0, 2.5)
    A_a = U((W, 1, 3), 0.0, 3.0)
    f_w = U((W, 1, 3), 0.2, 3.0)
    f_a = U((W, 1, 3), 0.2, 3.0)
    p_w = U((W, 1, 3), 0.0, 2 * math.pi)
    p_a = U((W, 1, 3), 0.0, 2 * math.pi)
    b_g = 0.01 * Nrm((W, 1, 3))
    b_a = 0.05 * Nrm((W, 1, 3))
    noise_on = torch.ones((W, 1, 1), **f64)
    if edge_cases:
        quiet = (sel >= 0.02) & (sel < 0.07)
        A_w[quiet] = U((int(quiet.sum()), 1, 3), 0.0, 0.005)
        noise_on[quiet] = 0.0
    tt = t.unsqueeze(-1)
    w = A_w * torch.sin(2 * math.pi * f_w * tt + p_w) + b_g + noise_on * (SIGMA_G / math.sqrt(dt)) * Nrm((W, N + 1, 3))
    a = A_a * torch.sin(2 * math.pi * f_a * tt + p_a) + b_a + noise_on * (SIGMA_A / math.sqrt(dt)) * Nrm((W, N + 1, 3))
    a[..., 2] += GRAV[2]
    knots = torch.cat([tt, w, a], dim=-1).contiguous()

    lin = torch.cat([b_g.squeeze(1) + noise_on.squeeze(1) * 1e-3 * Nrm((W, 3)),
                     b_a.squeeze(1) + 1e-2 * Nrm((W, 3))], dim=1).contiguous()
    q = Nrm((W, 4))
    q = q / q.norm(dim=1, keepdim=True)
    q = torch.where(q[:, 3:4] < 0, -q, q).contiguous()
    return knots, lin, q


def make_stream(W, N, seed=BASE_SEED, rate=200.0, device="cpu", phase=0.0):
    """ONE IMU stream of W * N + 1 readings at `rate` (same signal model as make_windows) and W update times, one every N
    samples: stream [K, 7], update_times [W], lin [W, 6], q_k_lin [W, 4].  phase = 0: the update times fall ON the IMU grid
    (every window has exactly N whole intervals, no tail -- the shape BASELINE's "N-sample windows" are quoted on);
    0 < phase < 1: that fraction of a sample period later (N whole intervals + the partial tail interval of
    GraphSolver_IMU.cpp:64-69, the first window excepted)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    f64 = dict(dtype=torch.float64, device=device)
    K = W * N + 1
    dt = 1.0 / rate
    t = 1275.1 + torch.arange(K, **f64) / rate
    U = lambda lo, hi: lo + (hi - lo) * torch.rand((1, 3), generator=g, **f64)
    A_w, A_a, f_w, f_a = U(0.0, 2.5), U(0.0, 3.0), U(0.2, 3.0), U(0.2, 3.0)
    p_w, p_a = U(0.0, 2 * math.pi), U(0.0, 2 * math.pi)
    b_g = 0.01 * torch.randn((1, 3), generator=g, **f64)
    b_a = 0.05 * torch.randn((1, 3), generator=g, **f64)
    stream = torch.empty((K, 7), **f64)
    stream[:, 0] = t
    tt = t.unsqueeze(-1)
    stream[:, 1:4] = A_w * torch.sin(2 * math.pi * f_w * tt + p_w) + b_g + (SIGMA_G / math.sqrt(dt)) * torch.randn((K, 3), generator=g, **f64)
    stream[:, 4:7] = A_a * torch.sin(2 * math.pi * f_a * tt + p_a) + b_a + (SIGMA_A / math.sqrt(dt)) * torch.randn((K, 3), generator=g, **f64)
    stream[:, 6] += GRAV[2]
    del tt
    update = t[N::N].clone() + phase * dt
    lin = torch.cat([b_g + 1e-3 * torch.randn((W, 3), generator=g, **f64), b_a + 1e-2 * torch.randn((W, 3), generator=g, **f64)], dim=1).contiguous()
    q = torch.randn((W, 4), generator=g, **f64)
    q = q / q.norm(dim=1, keepdim=True)
    q = torch.where(q[:, 3:4] < 0, -q, q).contiguous()
    return stream, update.contiguous(), lin, q


def make_states(out_alpha, out_beta, out_q, out_DT, lin, model, seed=BASE_SEED + 7, device="cpu",
                grav=GRAV):
    """State pairs for the evaluateError sweep (SURVEY.md 8(d) cfg 4): state_i random with biases
    near the linearisation point, state_j = predicted state (GraphSolver_IMU.cpp:263-307) perturbed
    by N(0, [1e-3 rad, 1e-4, 1e-2 m/s, 1e-3, 1e-2 m]).  Returns xi, xj [F,16]."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    f64 = dict(dtype=torch.float64, device=device)
    F = lin.shape[0]

    def Nrm(shape):
        return torch.randn(shape, generator=g, **f64)

    qi = Nrm((F, 4)); qi = qi / qi.norm(dim=1, keepdim=True)
    qi = torch.where(qi[:, 3:4] < 0, -qi, qi)
    bg = lin[:, 0:3] + 1e-3 * Nrm((F, 3))
    ba = lin[:, 3:6] + 1e-2 * Nrm((F, 3))
    v = 2.0 * Nrm((F, 3))
    p = 10.0 * Nrm((F, 3))
    xi = torch.cat([qi, bg, v, ba, p], dim=1)

    def quat_mul(q, p_):  # JPL product (quat_ops.h:115-128), batched
        qv, qw = q[:, :3], q[:, 3:4]
        pv, pw = p_[:, :3], p_[:, 3:4]
        ov = qw * pv + pw * qv - torch.cross(qv, pv, dim=1)
        ow = qw * pw - (qv * pv).sum(1, keepdim=True)
        o = torch.cat([ov, ow], dim=1)
        o = torch.where(o[:, 3:4] < 0, -o, o)
        return o / o.norm(dim=1, keepdim=True)

    def rot(q):  # quat_2_Rot (quat_ops.h:104-109), batched
        x, y, z, w_ = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        c = 2 * w_ * w_ - 1
        R = torch.zeros((q.shape[0], 3, 3), **f64)
        sk = torch.zeros_like(R)
        sk[:, 0, 1], sk[:, 0, 2] = -z, y
        sk[:, 1, 0], sk[:, 1, 2] = z, -x
        sk[:, 2, 0], sk[:, 2, 1] = -y, x
        R = c[:, None, None] * torch.eye(3, **f64) - 2 * w_[:, None, None] * sk + 2 * q[:, :3, None] * q[:, None, :3]
        return R

    gvec = torch.tensor(grav, **f64)
    DT = out_DT.reshape(F, 1)
    RT = rot(qi).transpose(1, 2)              # quat_2_Rot(Inv(q)) = R^T
    qj = quat_mul(out_q, qi)
    rb = torch.bmm(RT, out_beta.unsqueeze(-1)).squeeze(-1)
    ra = torch.bmm(RT, out_alpha.unsqueeze(-1)).squeeze(-1)
    if model == 1:
        vj = v - gvec * DT + rb
        pj = p + v * DT - 0.5 * gvec * DT * DT + ra
    else:
        vj = v + rb
        pj = p + v * DT + ra
    # perturb
    dth = 1e-3 * Nrm((F, 3))
    dq = torch.cat([0.5 * dth, torch.ones((F, 1), **f64)], dim=1)
    dq = dq / dq.norm(dim=1, keepdim=True)
    qj = quat_mul(dq, qj)
    xj = torch.cat([qj, bg + 1e-4 * Nrm((F, 3)), vj + 1e-2 * Nrm((F, 3)), ba + 1e-3 * Nrm((F, 3)),
                    pj + 1e-2 * Nrm((F, 3))], dim=1)
    return xi.contiguous(), xj.contiguous()
