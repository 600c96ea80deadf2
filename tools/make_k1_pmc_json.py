#!/usr/bin/env python3
"""Builds profiles/<tag>_k1_pmc.json from the two PMC passes of tools/collect_profiles.sh.

usage: make_k1_pmc_json.py <tag> [images_per_launch (default: from gpurun_out/<tag>_bench.json)]
       [w=752] [h=480]
FETCH_SIZE / WRITE_SIZE are reported in KiB; their scale is calibrated on the 256 MiB
bitwise_not kernel bench.py runs under OKVFE_PMC_CALIB=1 in the same process (it reads and writes
exactly 262144 KiB): on gfx950 FETCH_SIZE comes out at half the bytes read, WRITE_SIZE at 1.0."""
import json
import os
import sys

tag = sys.argv[1]
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_map_free, _cand = False, 0.0
try:  # images per launch of the profiled command = the bench line of the same collection
    _bn = os.path.join(_root, "gpurun_out", f"{tag}_bench.json")
    if not os.path.exists(_bn):  # shape tags (round4_v3_hilti): the bench line of that workload
        _base, _, _shape = tag.rpartition("_")
        _bn = os.path.join(_root, "gpurun_out", f"{_base}_bench_{_shape}.json")
    _b = json.load(open(_bn))
    _default_n = int(_b["config"].get("cameras_per_multiframe", 2)) * int(_b["config"]["stereo_frames_per_launch"])
    _r = _b.get("roofline", {})
    _map_free = _r.get("live_bound") == "valu"  # the launch wrote no score map (round 4 default)
except Exception:
    _default_n = 512
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else _default_n
w = int(sys.argv[3]) if len(sys.argv) > 3 else 752
h = int(sys.argv[4]) if len(sys.argv) > 4 else 480
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")


def rows(counter):
    d = json.load(open(os.path.join(src, f"{tag}_pmc_{counter}.json")))
    # the pipeline's kernel (score + fused NMS), not the score-only launches bench.py adds after
    # the timed region
    name = next((k for k in d if k.startswith("harris_kernel") and "true" in k),
                next(k for k in d if k.startswith("harris_kernel")))
    k1 = d[name]
    cal = next(v for k, v in d.items() if "bitwise_not" in k)
    return name, k1["mean_per_dispatch"][counter], cal["mean_per_dispatch"][counter]


if _map_free:  # candidates per image: algorithmic bytes of the bench line = (P + 12 C) images
    try:
        _cand = (_r["algorithmic_bytes_per_launch"] / float(_default_n) - w * h) / 12.0
    except Exception:
        _cand = 0.0
valu_insts = None
try:  # vector-ALU instructions of the same kernel (the SQ pass of the same collection)
    _sq = json.load(open(os.path.join(src, f"{tag}_pmc_sq.json")))
    _k = next((k for k in _sq if k.startswith("harris_kernel") and "true" in k), None)
    valu_insts = _sq[_k]["mean_per_dispatch"]["SQ_INSTS_VALU"] if _k else None
except Exception:
    pass
name, fetch, fetch_cal = rows("FETCH_SIZE")
_, write, write_cal = rows("WRITE_SIZE")
CAL_KIB = 262144.0
f_scale, w_scale = CAL_KIB / fetch_cal, CAL_KIB / write_cal
rd, wr = fetch * f_scale * 1024.0, write * w_scale * 1024.0
# second calibration, on K1's own access shapes (tools/ubench/fetch_calib: 1 GiB per kernel)
calib_shapes = None
try:
    cf = json.load(open(os.path.join(src, f"{tag}_calib_FETCH_SIZE.json")))
    cw = json.load(open(os.path.join(src, f"{tag}_calib_WRITE_SIZE.json")))
    gib = 1048576.0
    calib_shapes = {"read4_fetch_scale": gib / cf["read4"]["mean_per_dispatch"]["FETCH_SIZE"],
                    "read16_fetch_scale": gib / cf["read16"]["mean_per_dispatch"]["FETCH_SIZE"],
                    "write16_write_scale": gib / cw["write16"]["mean_per_dispatch"]["WRITE_SIZE"],
                    "note": "1 GiB per kernel; read4 = one dword per lane (K1's loads), write16 = 16 B "
                            "per lane (K1's stores): the same scales as the bitwise_not calibration"}
except Exception:
    pass
out = {
    "kernel": name,
    "workload": f"{w}x{h} u8 x {n_img} images per launch (bench.py --lanes 1)",
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- "
               "python bench.py --steps 3 --warmup 3 --no-cpu-baseline --lanes 1 "
               "(two separate passes, tools/collect_profiles.sh)",
    "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
    "calibration": {"kernel": "at::native bitwise_not, 262144 KiB read + 262144 KiB written",
                    "FETCH_SIZE_KiB": fetch_cal, "WRITE_SIZE_KiB": write_cal,
                    "fetch_scale": f_scale, "write_scale": w_scale},
    "hbm_read_bytes": rd, "hbm_write_bytes": wr,
    "hbm_bytes_per_launch": rd + wr,
    "images_per_launch": n_img,
    "hbm_bytes_per_image": (rd + wr) / n_img,
    "width": w, "height": h, "map_free": _map_free, "candidates_per_image": _cand,
    "algorithmic_bytes_per_launch": ((w * h + 12.0 * _cand) if _map_free else 5 * w * h) * n_img,
    "ratio_to_algorithmic": (rd + wr) / (((w * h + 12.0 * _cand) if _map_free else 5.0 * w * h) * n_img),
    "valu_insts_per_image": (valu_insts / n_img) if valu_insts else None,
    "read_ratio": rd / (1.0 * w * h * n_img), "write_ratio": wr / (4.0 * w * h * n_img),
    "calibration_on_k1_access_shapes": calib_shapes,
}
dst = os.path.join(root, "profiles", f"{tag}_k1_pmc.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
