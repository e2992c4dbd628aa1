"""Global localisation as the reference runs it (threadLocalization::globalization,
src/localization/localization.cpp:483-540): candidate poses that pass CVPmap::isValidPose are scored against the
observed scan, the k best (topK, :126-149) seed the particle filter and the best one is the pose estimate.

The reference draws its candidates with an unseeded `rand()` (randomPose, :88-103: x = (rand() % 1000) / 1000 * range,
theta = (rand() % 360) degrees) one at a time until 10 000 valid ones are scored; here the caller passes any pose
list (random_poses() restates the reference's lattice) and everything runs on the device behind the C ABI.
"""
import numpy as np
import torch

from . import capi

N_GLOBAL = 10000      # g_num_of_global_particles, globaldef.cpp:15
N_KEEP = 60           # g_num_of_particles


def random_poses(n, x_range, y_range, rng):
    """randomPose's lattice: x, y on a 1/1000 grid of the ranges, theta in whole degrees (PM_D2R = 3.141592654 / 180)."""
    d2r = np.float64(3.141592654) / 180.0
    p = np.stack([rng.integers(0, 1000, n) * 0.001 * x_range, rng.integers(0, 1000, n) * 0.001 * y_range,
                  rng.integers(0, 360, n) * d2r], 1)
    return p.astype(np.float32)


def global_search(field, d_prob, d_poses, scan, scorer="likelihood", k=N_KEEP, free_thr=0.3, unknown=0.5):
    """d_poses: [P][3] candidates (device).  scan: dict with the observation in the form the scorer needs -
    "likelihood": field built (field.build / build_from_grid), scan["x"], scan["y"] device floats (scan frame, m; CVPmap's
    frame, or the plain pose (+) point frame with scan["std_frame"] = True);
    "beam": scan["r"] (m), scan["bad"] (uint8) device arrays of n beams;
    "icp": scan["x"], scan["y"] device floats of n beams (CVPmap::obsLikelyhood, the reference's choice).
    Returns (best pose[3], indices of the k best among the valid candidates' original positions, their scores)."""
    ctx = field.ctx
    P = d_poses.shape[0]
    d_valid = torch.zeros(P, dtype=torch.uint8, device=d_poses.device)
    capi.valid_poses(field, d_prob, d_poses, P, d_valid, free_thr, unknown)
    ctx.sync()
    keep = torch.nonzero(d_valid).flatten()
    if keep.numel() == 0:
        raise ValueError("no valid candidate pose")
    d_cand = d_poses[keep].contiguous()
    M = d_cand.shape[0]
    d_score = torch.zeros(M, dtype=torch.float32, device=d_poses.device)
    # the gather and the fill above run on torch's current stream, the scorer on the context's: order them explicitly
    # (the context's own stream is a blocking one and would wait for the default stream anyway; a caller-owned stream
    # set with ctx.set_stream might not)
    torch.cuda.current_stream().synchronize()
    if scorer == "likelihood":
        field.score_poses(scan["x"], scan["y"], scan["x"].numel(), scan.get("decimation", 3), d_cand, M, d_score,
                          std_frame=bool(scan.get("std_frame", False)))
    elif scorer == "beam":
        capi.beam_model_scores(field, d_prob, d_cand, M, scan["r"], scan["bad"], scan["r"].numel(), d_score)
    elif scorer == "icp":
        capi.icp_goodness_scores(field, d_prob, d_cand, M, scan["x"], scan["y"], scan["x"].numel(), d_score)
    else:
        raise ValueError("scorer: likelihood | beam | icp")
    ctx.sync()
    idx, sc = capi.pose_topk(ctx, d_score, M, k)
    orig = keep.cpu().numpy()[idx]
    best = d_poses[int(orig[0])].cpu().numpy()
    return best, orig, sc
