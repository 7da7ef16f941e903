"""Second opinions for the oracle's restatements of the SURVEY 8 f3 trackers' filters: numpy restatements written from the
REFERENCE's sources — not from oracle/ — in float32 (the reference's own precision, numpy's operation order) and float64 (what the
formulas mean), emitting the fixture tests/golden/f3_second_opinions.npz that tests/test_oracle_second_opinions.py checks oracle/ against.
The reference holds no test vector for any of these (its tests/ stop at N <= 3 assignments, IoU pairs, the XYSR filter and SORT ids), so
this is the independent check the oracle's last-bit choices are bounded by: where the two float32 evaluations differ it is by summation
order only, and the float64 column says which digits are the formula's.

  python tests/golden/make_f3_second_opinions.py        (build container only; needs numpy, nothing of the repository)

Restated here, with the reference lines they follow:
  * HybridSORT's nine-state filter: F, H, Q, R, P0 (src/trackers/hybridsort.cpp:26-57), init (:59-64), predict (:66-69) with the
    tracker's guard ds + s <= 0 -> ds = 0 (:257-259), update with S^-1 (:71-88), the measurement [u, v, s, c, r] of a box (:181-193), the
    all-zero measurement of an unmatched track (:315-320).
  * UCMCTrack's ground-plane filter in double precision: image-space mapping (src/trackers/ucmc.cpp:123-139), birth (:146-190), predict
    (:31-34), the Joseph-form update (:36-52), and the association cost d' S^-1 d + log det S (:202-213).
  * The XYAH motion gate StrongSORT blends into its appearance cost: project (src/motion/kalman_filter.cpp:60-75 with the XYAH noise of
    kalman_filters/xyah_kf.cpp:50-62), gating_distance's "maha" branch as written — |S^-1 d|^2, the squared norm of chol.solve(d)
    (:148-176) — and gate_cost_matrix (src/trackers/strongsort.cpp:449-492).
(BoostTrack's confidence boosts, src/trackers/boosttrack.cpp:360-423, are max / pow expressions of IoU values the oracle computes with the pinned
iou_batch; they have no precision choice to cross-check and are pinned by the hand-derived case of tests/test_oracle_known_answers.py.)
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


# ---- HybridSORT ------------------------------------------------------------------------------------------------------------
def hyb_mats(dt):
    F = np.eye(9, dtype=dt)
    F[0, 5] = F[1, 6] = F[2, 7] = F[3, 8] = 1
    H = np.zeros((5, 9), dt)
    for k in range(5):
        H[k, k] = 1
    Q = np.eye(9, dtype=dt) * dt(0.1)
    Q[8, 8] = Q[7, 7] = Q[5, 5] = Q[6, 6] = dt(0.01)
    R = np.eye(5, dtype=dt)
    R[2, 2] = dt(10.0)
    R[3, 3] = dt(0.01)
    P = np.eye(9, dtype=dt) * dt(10.0)
    P[5:9, 5:9] *= dt(1000.0)
    return F, H, Q, R, P


def box_to_z(b, dt):
    b = b.astype(dt)
    w, h = b[2] - b[0], b[3] - b[1]
    u, v = b[0] + w / dt(2), b[1] + h / dt(2)
    s = w * h
    r = w / h if h > dt(1e-6) else dt(0)
    return np.array([u, v, s, b[4], r], dt)


def hyb_replay(boxes_per_frame, dt):
    """boxes_per_frame[f] = [x1, y1, x2, y2, conf] of ONE object or None (no detection: the track takes the zero measurement).
    Returns per frame (x, P) after the frame, frame 0 = the birth."""
    F, H, Q, R, P0 = hyb_mats(dt)
    x = np.zeros(9, dt)
    x[:5] = box_to_z(boxes_per_frame[0], dt)
    P = P0.copy()
    out = [(x.copy(), P.copy())]
    I9 = np.eye(9, dtype=dt)
    for b in boxes_per_frame[1:]:
        if x[7] + x[2] <= 0:
            x[7] = 0
        x = F @ x
        P = F @ P @ F.T + Q
        z = box_to_z(b, dt) if b is not None else np.zeros(5, dt)
        S = H @ P @ H.T + R
        K = P @ H.T @ np.linalg.inv(S)
        x = x + K @ (z - H @ x)
        P = (I9 - K @ H) @ P
        out.append((x.copy(), P.copy()))
    return out


# ---- UCMCTrack -------------------------------------------------------------------------------------------------------------
def ucmc_map_image(cx, bottom, w, h):
    y = np.array([cx * 0.01, bottom * 0.01])
    ex = max(0.02, min(0.13, 0.0005 * w))
    ey = max(0.02, min(0.10, 0.0005 * h))
    return y, np.diag([ex * ex, ey * ey])


def ucmc_replay(boxes_per_frame, wx=5.0, wy=5.0, vmax=10.0, dt=1.0 / 30.0):
    """float32 boxes [x1, y1, x2, y2] of ONE object per frame; returns per frame (x, P), frame 0 = the birth."""
    def meas(b):
        b = np.asarray(b, np.float32)
        w, h = np.float32(b[2] - b[0]), np.float32(b[3] - b[1])
        cx = np.float32(b[0] + w / np.float32(2))
        return ucmc_map_image(float(cx), float(b[3]), float(w), float(h))
    F = np.eye(4)
    F[0, 1] = F[2, 3] = dt
    Hm = np.zeros((2, 4))
    Hm[0, 0] = Hm[1, 2] = 1
    G = np.array([[0.5 * dt * dt, 0], [dt, 0], [0, 0.5 * dt * dt], [0, dt]])
    Q = G @ np.diag([wx, wy]) @ G.T
    y, _ = meas(boxes_per_frame[0])
    x = np.array([y[0], 0.0, y[1], 0.0])
    P = np.diag([1.0, vmax * vmax / 3.0, 1.0, vmax * vmax / 3.0])
    out = [(x.copy(), P.copy())]
    for b in boxes_per_frame[1:]:
        x = F @ x
        P = F @ P @ F.T + Q
        y, R = meas(b)
        S = Hm @ P @ Hm.T + R
        K = P @ Hm.T @ np.linalg.inv(S)
        x = x + K @ (y - Hm @ x)
        IKH = np.eye(4) - K @ Hm
        P = IKH @ P @ IKH.T + K @ R @ K.T
        out.append((x.copy(), P.copy()))
    return out


def ucmc_distance(x, P, y, R):
    Hm = np.zeros((2, 4))
    Hm[0, 0] = Hm[1, 2] = 1
    d = y - Hm @ x
    S = Hm @ P @ Hm.T + R
    return float(d @ np.linalg.inv(S) @ d + np.log(np.linalg.det(S)))


# ---- XYAH gate (StrongSORT) ------------------------------------------------------------------------------------------------
def xyah_gate(mean, cov, meas, cost, lam, gated_cost, dt):
    mean, cov, meas, cost = mean.astype(dt), cov.astype(dt), meas.astype(dt), cost.astype(dt)
    h = mean[3]
    std = np.array([dt(1.0 / 20.0) * h, dt(1.0 / 20.0) * h, dt(1e-1), dt(1.0 / 20.0) * h], dt)
    S = cov[:4, :4] + np.diag(std * std)
    d = meas - mean[:4]
    g = np.array([np.sum(np.linalg.solve(S, di) ** 2) for di in d], dt)  # |S^-1 d|^2, as the reference writes it
    c = cost.copy()
    c[g > dt(9.4877)] = dt(gated_cost)
    return g, dt(lam) * c + (dt(1) - dt(lam)) * g


def main():
    r = np.random.default_rng(20260930)
    out = {}
    # four objects moving on straight lines, far apart (every association of the trackers is forced), confidences above every threshold;
    # object 3 is never detected again after frame 4 (HybridSORT: all-zero measurements from then on)
    Fm, K = 12, 4
    c0 = np.array([[200, 300], [700, 250], [1200, 600], [1600, 350]], np.float64)
    vel = np.array([[6, 2], [-5, 3], [4, -4], [-3, -2]], np.float64)
    wh = np.array([[60, 140], [80, 170], [70, 150], [50, 120]], np.float64)
    boxes = np.zeros((Fm, K, 5), np.float32)
    for f in range(Fm):
        c = c0 + vel * f + r.normal(0, 0.8, (K, 2))
        s = wh * (1.0 + 0.01 * f) + r.normal(0, 0.6, (K, 2))
        boxes[f, :, 0:2] = c - s / 2
        boxes[f, :, 2:4] = c + s / 2
        boxes[f, :, 4] = 0.9 - 0.01 * np.arange(K) - 0.002 * f
    present = np.ones((Fm, K), bool)
    present[5:, 3] = False
    out["boxes"] = boxes
    out["present"] = present
    for name, dt in (("f32", np.float32), ("f64", np.float64)):
        xs = np.zeros((Fm, K, 9), dt)
        Ps = np.zeros((Fm, K, 81), dt)
        for k in range(K):
            seq = [boxes[f, k] if present[f, k] else None for f in range(Fm)]
            for f, (x, P) in enumerate(hyb_replay(seq, dt)):
                xs[f, k], Ps[f, k] = x, P.reshape(-1)
        out[f"hyb_x_{name}"], out[f"hyb_P_{name}"] = xs, Ps
    # UCMC: the three objects that stay
    ux = np.zeros((Fm, 3, 4))
    uP = np.zeros((Fm, 3, 16))
    for k in range(3):
        for f, (x, P) in enumerate(ucmc_replay([boxes[f, k, :4] for f in range(Fm)])):
            ux[f, k], uP[f, k] = x, P.reshape(-1)
    out["ucmc_x"], out["ucmc_P"] = ux, uP
    n, m = 7, 9
    X = r.normal(0, 3, (n, 4))
    A = r.normal(0, 1, (n, 4, 4))
    Pm = A @ A.transpose(0, 2, 1) + 0.5 * np.eye(4)
    Y = r.normal(0, 3, (m, 2))
    Rm = np.stack([np.diag(r.uniform(0.02, 0.2, 2) ** 2) for _ in range(m)])
    out["ucmc_d_x"], out["ucmc_d_P"], out["ucmc_d_y"], out["ucmc_d_R"] = X, Pm, Y, Rm
    out["ucmc_d"] = np.array([[ucmc_distance(X[i], Pm[i], Y[j], Rm[j]) for j in range(m)] for i in range(n)])
    # XYAH gate: dense covariances (what a filter after a few updates holds) and measurements near and far
    n, m = 6, 8
    mean = np.zeros((n, 8), np.float32)
    mean[:, 0:2] = r.uniform(100, 900, (n, 2))
    mean[:, 2] = r.uniform(0.3, 0.6, n)
    mean[:, 3] = r.uniform(80, 200, n)
    mean[:, 4:] = r.normal(0, 1, (n, 4))
    A = r.normal(0, 1, (n, 8, 8))
    cov = (A @ A.transpose(0, 2, 1) * 4 + np.eye(8) * 10).astype(np.float32)
    meas = np.zeros((m, 4), np.float32)
    meas[:, 0:2] = mean[r.integers(0, n, m), 0:2] + r.normal(0, 12, (m, 2))
    meas[:, 2] = r.uniform(0.3, 0.6, m)
    meas[:, 3] = r.uniform(80, 200, m)
    cost = r.uniform(0, 0.4, (n, m)).astype(np.float32)
    out["gate_mean"], out["gate_cov"], out["gate_meas"], out["gate_cost_in"] = mean, cov.reshape(n, 64), meas, cost
    for name, dt in (("f32", np.float32), ("f64", np.float64)):
        G = np.zeros((n, m), dt)
        C2 = np.zeros((n, m), dt)
        for i in range(n):
            G[i], C2[i] = xyah_gate(mean[i], cov[i], meas, cost[i], 0.98, 1e5, dt)
        out[f"gate_g_{name}"], out[f"gate_c_{name}"] = G, C2
    np.savez_compressed(os.path.join(HERE, "f3_second_opinions.npz"), **out)
    print("wrote", os.path.join(HERE, "f3_second_opinions.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
