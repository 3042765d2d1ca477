"""Trajectory accuracy as the reference harness defines it (the yardstick of "ATE delta vs ref").

- alignment: Eigen::umeyama(estimate, ground truth, with scaling) over ground-truth positions that are
  linearly interpolated to the estimate's timestamps
  (reference framework/shared/src/outputs/TrajectoryAlignmentMethod.cpp:85-156);
- ATE: every aligned estimate is compared with the first ground-truth pose whose timestamp is not
  earlier; Euclidean position error; mean / max / RMSE
  (reference framework/shared/src/metrics/ATEMetric.cpp:74-151), plus the orientation error of the last pose;
- RPE: RMSE over consecutive estimates of the translation of (E_i E_j^-1)(G_i G_j^-1)^-1
  (reference framework/shared/src/metrics/RPEMetric.cpp:49-143).
"""
import numpy as np


def umeyama(src, dst, with_scaling=True):
    """Least-squares similarity transform dst ~ c * R * src + t (Umeyama 1991). src, dst: [3, N]."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    n = src.shape[1]
    mu_s, mu_d = src.mean(1, keepdims=True), dst.mean(1, keepdims=True)
    s0, d0 = src - mu_s, dst - mu_d
    var_s = (s0 * s0).sum() / n
    sigma = d0 @ s0.T / n
    U, D, Vt = np.linalg.svd(sigma)
    S = np.ones(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2] = -1.0
    R = U @ np.diag(S) @ Vt
    c = (D * S).sum() / var_s if (with_scaling and var_s > 0) else 1.0
    T = np.eye(4)
    T[:3, :3] = c * R
    T[:3, 3] = (mu_d - c * R @ mu_s)[:, 0]
    return T


def associate_positions(gt_poses, gt_ts, est_ts):
    """Ground-truth positions interpolated to the estimate's timestamps (clamped at the ends)."""
    gt_pos = np.asarray(gt_poses, np.float64)[:, :3, 3]
    gt_ts = np.asarray(gt_ts, np.float64)
    out = np.empty((len(est_ts), 3))
    for i, t in enumerate(np.asarray(est_ts, np.float64)):
        g = int(np.searchsorted(gt_ts, t, side="left"))    # first gt with ts >= t
        if g == 0:
            out[i] = gt_pos[0]
        elif g == len(gt_ts):
            out[i] = gt_pos[-1]
        elif gt_ts[g] == t:
            out[i] = gt_pos[g]
        else:
            a = (t - gt_ts[g - 1]) / (gt_ts[g] - gt_ts[g - 1])
            out[i] = gt_pos[g - 1] * (1 - a) + gt_pos[g] * a
    return out


def align_umeyama(est_poses, gt_poses, est_ts, gt_ts):
    est_pos = np.asarray(est_poses, np.float64)[:, :3, 3]
    gt_pos = associate_positions(gt_poses, gt_ts, est_ts)
    if len(est_pos) < 3 or np.allclose(est_pos.var(0).sum(), 0.0):
        # degenerate (all estimates identical, e.g. the first frames): translation only
        T = np.eye(4)
        T[:3, 3] = gt_pos.mean(0) - est_pos.mean(0)
        return T
    return umeyama(est_pos.T, gt_pos.T, True)


def ate(est_poses, gt_poses, est_ts, gt_ts, align=True):
    """Returns dict(rmse, mean, max, last, per_frame) in metres."""
    est_poses = np.asarray(est_poses, np.float64)
    gt_poses = np.asarray(gt_poses, np.float64)
    est_ts = np.asarray(est_ts, np.float64)
    gt_ts = np.asarray(gt_ts, np.float64)
    T = align_umeyama(est_poses, gt_poses, est_ts, gt_ts) if align else np.eye(4)
    errs = []
    g = 0
    for i in range(len(est_poses)):
        while g < len(gt_ts) and gt_ts[g] < est_ts[i]:
            g += 1
        if g == len(gt_ts):
            break
        aligned = T @ est_poses[i]
        d = np.abs(gt_poses[g][:3, 3] - aligned[:3, 3])
        errs.append(float(np.sqrt((d * d).sum())))
    errs = np.asarray(errs)
    if errs.size == 0:
        return {"rmse": float("nan"), "mean": float("nan"), "max": float("nan"), "last": float("nan"),
                "per_frame": errs}
    return {"rmse": float(np.sqrt((errs * errs).mean())), "mean": float(errs.mean()), "max": float(errs.max()),
            "last": float(errs[-1]), "per_frame": errs}


def _associate_not_earlier(est_ts, gt_ts):
    """Index of the first ground-truth sample whose timestamp is not earlier than each estimate's (the
    association of ATEMetric.cpp:88-118 and RPEMetric.cpp:81-107); -1 once the ground truth has run out."""
    out = np.full(len(est_ts), -1, int)
    g = 0
    for i, t in enumerate(est_ts):
        while g < len(gt_ts) and gt_ts[g] < t:
            g += 1
        if g == len(gt_ts):
            break
        out[i] = g
    return out


def orientation_error_deg(gt_pose, aligned_pose):
    """Angle of gt_R^T * aligned_R after removing the alignment's scale (ATEMetric.cpp:132-139)."""
    R = np.asarray(aligned_pose, np.float64)[:3, :3]
    R = R / np.linalg.norm(R[:, 0])
    D = np.asarray(gt_pose, np.float64)[:3, :3].T @ R
    c = (np.trace(D) - 1.0) / 2.0
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0))))


def rpe(aligned_poses, gt_poses, est_ts, gt_ts):
    """RPE_RMSE over the (aligned) estimates so far; 0 with fewer than two associated poses."""
    aligned_poses = np.asarray(aligned_poses, np.float64)
    gt_poses = np.asarray(gt_poses, np.float64)
    assoc = _associate_not_earlier(np.asarray(est_ts, np.float64), np.asarray(gt_ts, np.float64))
    norms, count = 0.0, 0
    prev = None
    for i, g in enumerate(assoc):
        if g < 0:
            break
        if prev is not None:
            pi, pg = prev
            gt_diff = gt_poses[pg] @ np.linalg.inv(gt_poses[g])
            es_diff = aligned_poses[pi] @ np.linalg.inv(aligned_poses[i])
            E = es_diff @ np.linalg.inv(gt_diff)
            norms += float(E[:3, 3] @ E[:3, 3])
            count += 1
        prev = (i, g)
    return float(np.sqrt(norms / count)) if count else 0.0


def frame_metrics(est_poses, gt_poses, est_ts, gt_ts):
    """The per-frame accuracy columns of a SLAMBench log, the way the harness produces them: after every
    frame the trajectory so far is re-aligned (Umeyama with scale) and ATE / RPE are evaluated over it.
    Returns a list of dicts with ATE_RMSE, MaxATE, MeanATE, OrientationError, RPE_RMSE."""
    est_poses = np.asarray(est_poses, np.float64)
    gt_poses = np.asarray(gt_poses, np.float64)
    est_ts = np.asarray(est_ts, np.float64)
    gt_ts = np.asarray(gt_ts, np.float64)
    rows = []
    for n in range(1, len(est_poses) + 1):
        T = align_umeyama(est_poses[:n], gt_poses, est_ts[:n], gt_ts)
        aligned = np.einsum("ij,njk->nik", T, est_poses[:n])
        a = ate(est_poses[:n], gt_poses, est_ts[:n], gt_ts, align=True)
        assoc = _associate_not_earlier(est_ts[:n], gt_ts)
        last = max((i for i in range(n) if assoc[i] >= 0), default=None)
        rows.append({
            "ATE_RMSE": a["rmse"], "MaxATE": a["max"], "MeanATE": a["mean"],
            "OrientationError": orientation_error_deg(gt_poses[assoc[last]], aligned[last]) if last is not None else float("nan"),
            "RPE_RMSE": rpe(aligned, gt_poses, est_ts[:n], gt_ts),
        })
    return rows
