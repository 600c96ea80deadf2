"""Synthetic map for the matchToMap tests: old keyframe poses on an arc, 3-D landmarks in front of
them, each with several observations (descriptor = the landmark's base descriptor with a few bits
flipped, back-projection = the ray in the observing camera), and a current frame whose keypoints
are the landmark projections.  Test infrastructure only."""
import numpy as np

from okvis2_amd import synth


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def make_map(n_landmarks=6000, n_poses=12, seed=1, voc=None):
    rng = np.random.default_rng(seed)
    cam = synth.euroc_config().cams[0]
    poses = []
    for i in range(n_poses):
        a = 0.08 * (i - n_poses / 2)
        poses.append((rot_y(a).reshape(-1).copy(), np.array([0.25 * i - 1.5, 0.02 * i, 0.1 * np.sin(i)])))
    T1 = (rot_y(0.03).reshape(-1).copy(), np.array([0.4, 0.0, 0.3]))
    # landmarks: most in front of the rig (2..12 m), some behind / far to the side (FoV rejects)
    p = np.stack([rng.uniform(-6, 6, n_landmarks), rng.uniform(-3, 3, n_landmarks),
                  rng.uniform(1.5, 12, n_landmarks)], axis=1)
    p[rng.random(n_landmarks) < 0.08, 2] *= -1.0
    w4 = rng.choice([1.0, 2.0, -1.0, 0.5], n_landmarks, p=[0.7, 0.1, 0.1, 0.1])
    hp = np.concatenate([p * w4[:, None], w4[:, None]], axis=1)
    quality = rng.choice([1.0, 0.3, 0.05, 0.001], n_landmarks, p=[0.4, 0.3, 0.2, 0.1])
    n_obs = rng.integers(0, 7, n_landmarks)
    obs_begin = np.concatenate([[0], np.cumsum(n_obs)]).astype(np.int32)
    if voc is None:
        base = rng.integers(0, 256, (n_landmarks, 48), dtype=np.uint8)
    else:
        base = voc[rng.integers(0, len(voc), n_landmarks)] ^ rng.integers(0, 256, (n_landmarks, 48), dtype=np.uint8)
    obs_pose, obs_desc, obs_bp = [], [], []
    for l in range(n_landmarks):
        for _ in range(n_obs[l]):
            pi = int(rng.integers(0, n_poses))
            C = poses[pi][0].reshape(3, 3)
            ray = C.T @ (p[l] - poses[pi][1])
            obs_pose.append(pi)
            obs_bp.append(ray * rng.uniform(0.2, 3.0) / max(np.linalg.norm(ray), 1e-9))
            flip = (rng.random(48) < 0.04) * rng.integers(1, 256, 48)
            obs_desc.append(base[l] ^ flip.astype(np.uint8))
    obs_pose = np.array(obs_pose, dtype=np.int32)
    obs_desc = np.array(obs_desc, dtype=np.uint8).reshape(-1, 48)
    obs_bp = np.array(obs_bp, dtype=np.float64).reshape(-1, 3)
    return dict(cam=cam, poses=poses, T1=T1, hp=hp, quality=quality, obs_begin=obs_begin, obs_pose=obs_pose,
                obs_desc=obs_desc, obs_bp=obs_bp, base=base, p=p)


def make_frame(m, oracle, n_kps=700, seed=2):
    """Current-frame keypoints at the projections of (a subset of) the landmarks, descriptors close
    to the landmark's base descriptor."""
    rng = np.random.default_rng(seed)
    cam, (C1, r1) = m["cam"], m["T1"]
    C1 = C1.reshape(3, 3)
    kps = np.zeros(n_kps, dtype=oracle.KEYPOINT_DTYPE)
    desc = rng.integers(0, 256, (n_kps, 48), dtype=np.uint8)
    order = rng.permutation(len(m["p"]))
    k = 0
    for l in order:
        if k >= n_kps:
            break
        pc = C1.T @ (m["p"][l] - r1)
        st, pt, _ = oracle.cam_project(cam, pc)
        if st != 0:
            continue
        kps[k]["x"], kps[k]["y"] = pt[0] + rng.normal(0, 1.5), pt[1] + rng.normal(0, 1.5)
        kps[k]["size"] = 12.0
        flip = (rng.random(48) < 0.05) * rng.integers(1, 256, 48)
        desc[k] = m["base"][l] ^ flip.astype(np.uint8)
        k += 1
    kps[k:]["x"] = rng.uniform(0, cam.w, n_kps - k)
    kps[k:]["y"] = rng.uniform(0, cam.h, n_kps - k)
    kps[k:]["size"] = 12.0
    use = (rng.random(n_kps) < 0.9).astype(np.uint8)
    return kps, desc, use


def packed_set(pool, obs_desc, want):
    """The landmarks with status == want as (index, projections, desc_begin, pool rows)."""
    idx = np.flatnonzero(pool["status"] == want)
    begin = np.concatenate([[0], np.cumsum(pool["n_desc"][idx])]).astype(np.int32)
    rows = [obs_desc[pool["obs_rows"][l, r]] for l in idx for r in range(pool["n_desc"][l])]
    rows = np.array(rows, dtype=np.uint8).reshape(-1, 48)
    return idx, pool["projection"][idx], begin, rows
