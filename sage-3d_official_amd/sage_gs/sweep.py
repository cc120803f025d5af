"""Pose-file sweep driver: the frame-generation half of
`Code/data_pipeline/training_data_construction/generate_images.py` without Isaac Sim.

Input : `action_groundtruth.json` — `groundtruth_data[].{trajectory_id, instruction_index,
        sampled_points[].{point_id, position, rotation}}` (generate_actions.py:586-592;
        generate_images.py:180-227).
Work  : per trajectory, eye height forced to 1.2 m and the stored rotation passed as the Isaac orientation
        (generate_images.py:417-421), all frames of a trajectory rendered as ONE batch on the GPU.
Output: `<out>/trajectory_<id>/<scene>_<traj>_<idx:03d>.jpg` and `<out>/image_metadata.json` with the
        reference's fields (generate_images.py:414,572-609); existing trajectories are skipped unless
        --force (the reference's file-existence resume, :229-286).

    python -m sage_gs.sweep --scene scene.ply --actions action_groundtruth.json --scene-id 0001 --out frames/
"""
from __future__ import annotations

import argparse
import json
import os
from typing import Dict, List

import numpy as np

from . import camera as cam_conv

CAMERA_RESOLUTION = (1024, 768)          # generate_images.py:43
CAMERA_FOCAL_LENGTH = 8.0                # :44
CAMERA_HEIGHT = 1.2                      # :45


def load_trajectories(path) -> List[Dict]:
    """[{trajectory_id, instruction_index, points:[{point, position, rotation}]}], one entry per trajectory
    (the reference renders a trajectory once and shares the frames between its instructions)."""
    data = json.load(open(path, "r", encoding="utf-8"))
    seen, out = set(), []
    for item in data.get("groundtruth_data", []):
        tid = str(item["trajectory_id"])
        if tid in seen:
            continue
        seen.add(tid)
        pts = [{"point": sp["point_id"], "position": sp["position"], "rotation": sp["rotation"]}
               for sp in item.get("sampled_points", [])]
        out.append({"trajectory_id": tid, "instruction_index": item.get("instruction_index", 0), "points": pts})
    return out


def cameras_for(points, resolution=CAMERA_RESOLUTION):
    cams = []
    for p in points:
        pos, orient = cam_conv.datagen_pose(p)
        cams.append(cam_conv.reference_camera(resolution[0], resolution[1], pos, orient))
    return cams


def run(renderer, scene, trajectories, scene_id, out_dir, resolution=CAMERA_RESOLUTION, force=False, quality=95,
        chunk=64, on_frame=None):
    """Renders every trajectory (one GPU batch per `chunk` poses) and writes the reference's output layout.
    on_frame(trajectory_id, index, rgb uint8 [H,W,3]) — optional — sees each frame as it is handed to the JPEG encoder
    (the array `cam.get_rgba()[:, :, :3]` would be in generate_images.py:428-432)."""
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    sequences, total = [], 0
    for tr in trajectories:
        tdir = os.path.join(out_dir, f"trajectory_{tr['trajectory_id']}")
        names = [f"{scene_id}_{tr['trajectory_id']}_{i:03d}.jpg" for i in range(len(tr["points"]))]
        done = os.path.isdir(tdir) and all(os.path.exists(os.path.join(tdir, n)) for n in names)
        if not done or force:
            os.makedirs(tdir, exist_ok=True)
            cams = cameras_for(tr["points"], resolution)
            for c0 in range(0, len(cams), chunk):
                frames = renderer.render_batch(cams[c0:c0 + chunk], scene)            # [B,H,W,3] on the GPU
                # ONE pack and ONE device-to-host copy per chunk (the batch is contiguous: B stacked images are one tall image)
                b, h, w = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
                host = renderer.pack_rgba8(frames.reshape(b * h, w, 3)).cpu().numpy().reshape(b, h, w, 4)
                for k in range(b):
                    rgba = host[k]
                    if on_frame is not None:
                        on_frame(tr["trajectory_id"], c0 + k, rgba[:, :, :3])
                    Image.fromarray(rgba[:, :, :3]).save(os.path.join(tdir, names[c0 + k]), quality=quality)
        total += len(names)
        sequences.append({"scene_id": scene_id, "trajectory_id": tr["trajectory_id"],
                          "instruction_index": tr["instruction_index"], "frame_filenames": names,
                          "trajectory_sampled_points": [{"point_id": p["point"], "position": p["position"],
                                                         "rotation": p["rotation"]} for p in tr["points"]],
                          "sampling_info": {"sampled_points_count": len(names), "generated_images_count": len(names),
                                            "data_source": "sage_gs.sweep"}})
    meta = {"scene_id": scene_id, "scene_name": scene_id, "total_image_sequences": len(sequences),
            "frames_per_sequence": "variable_based_on_action_sampling", "image_resolution": list(resolution),
            "camera_settings": {"focal_length": CAMERA_FOCAL_LENGTH, "height": CAMERA_HEIGHT},
            "sequences": sequences, "processing_mode": {"type": "gpu_batch", "scene_reuse": True, "single_camera": True}}
    json.dump(meta, open(os.path.join(out_dir, "image_metadata.json"), "w", encoding="utf-8"), ensure_ascii=False, indent=2)
    return total


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--scene", required=True, help="3DGS .ply (standard layout) or PlayCanvas compressed .ply")
    ap.add_argument("--compressed", action="store_true")
    ap.add_argument("--actions", required=True, help="action_groundtruth.json")
    ap.add_argument("--scene-id", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--width", type=int, default=CAMERA_RESOLUTION[0])
    ap.add_argument("--height", type=int, default=CAMERA_RESOLUTION[1])
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args(argv)
    from . import ply, scenes
    from .renderer import Renderer
    arrays = (ply.load_compressed_ply if a.compressed else ply.load_ply)(a.scene)
    r = Renderer(a.device)
    scene = r.upload(ply.to_gaussians(arrays, a.device, scenes.MODEL_TO_WORLD))      # template.usda:120
    n = run(r, scene, load_trajectories(a.actions), a.scene_id, a.out, (a.width, a.height), a.force)
    print(f"[sage_gs.sweep] {n} frames -> {a.out}")


if __name__ == "__main__":
    main()
