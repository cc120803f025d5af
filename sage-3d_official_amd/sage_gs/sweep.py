"""Pose-file sweep driver: the frame-generation half of
`Code/data_pipeline/training_data_construction/generate_images.py` without Isaac Sim.

Input : `action_groundtruth.json` — `groundtruth_data[].{trajectory_id, instruction_index,
        sampled_points[].{point_id, position, rotation}}` (generate_actions.py:586-592;
        generate_images.py:180-227).
Work  : per trajectory, eye height forced to 1.2 m and the stored rotation passed as the Isaac orientation
        (generate_images.py:417-421), all frames of a trajectory rendered as ONE batch on the GPU.
Output: `<out>/images/trajectory_<id>/<scene>_<traj>_<idx:03d>.jpg` and `<out>/image_metadata.json` with the
        reference's fields (generate_images.py:308-309,395,414,572-609 — the file set a recorded run of the reference wrote is
        tests/golden/isaac_call_trace.json `files_written`); existing trajectories are skipped unless --force (the reference's
        file-existence resume, :229-286).

    python -m sage_gs.sweep --scene scene.ply --actions action_groundtruth.json --scene-id 0001 --out frames/

Many scenes, several processes (the reference's only parallelism, generate_images.py:72-75,136-139: `--instance-id` /
`--total-instances`, scenes dealt to instances by a hash of the scene id):

    python -m sage_gs.sweep --action-root actions/ --scene-root scenes/ --out frames/ --instance-id 3 --total-instances 8

renders the scenes `<action-root>/<scene_id>/action_groundtruth.json` whose stable hash falls to this instance — one process per
GPU (`--device auto`: cuda:LOCAL_RANK, else cuda:(instance_id mod device count)), scene replicated nowhere, no collective: the
embarrassingly parallel 8-GPU mode of a 1 000-scene data-generation run.  The hash is CRC-32 of the scene id: the reference's
`hash(scene_id)` is salted per process (PYTHONHASHSEED), so its instances disagree about who owns a scene; this one is the same
in every process, on every machine.
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


def scene_instance(scene_id: str, total_instances: int) -> int:
    """The instance (0 .. total_instances-1) that owns a scene: CRC-32 of its id, modulo the instance count.  Stable across
    processes and machines (unlike generate_images.py:137, whose hash(str) is salted per interpreter)."""
    import zlib
    if total_instances < 1:
        raise ValueError("total_instances must be >= 1")
    return zlib.crc32(str(scene_id).encode("utf-8")) % int(total_instances)


def scenes_of_instance(action_root, instance_id: int = 0, total_instances: int = 1) -> List[str]:
    """Scene ids (sub-directories of `action_root` holding an action_groundtruth.json, sorted as generate_images.py:133 sorts
    them) that fall to this instance."""
    if not 0 <= instance_id < total_instances:
        raise ValueError(f"instance_id({instance_id}) must be in range [0, {total_instances})")      # generate_images.py:91-92
    ids = sorted(d for d in os.listdir(action_root)
                 if os.path.isfile(os.path.join(action_root, d, "action_groundtruth.json")))
    return [i for i in ids if total_instances == 1 or scene_instance(i, total_instances) == instance_id]


def find_scene_file(scene_root, scene_id: str):
    """(path, compressed?) of a scene's Gaussians under scene_root: `<id>.ply`, `<id>/3dgs.ply`, or the PlayCanvas-compressed
    `<id>_compressed.ply` / `<id>/3dgs_compressed.ply` (README.md:210-231).  (A stage file `<id>.usda` is resolved by
    adapter.open_stage / parse_scene_usda, not here.)"""
    for rel, comp in ((f"{scene_id}.ply", False), (os.path.join(scene_id, "3dgs.ply"), False),
                      (f"{scene_id}_compressed.ply", True), (os.path.join(scene_id, "3dgs_compressed.ply"), True)):
        path = os.path.join(scene_root, rel)
        if os.path.isfile(path):
            return path, comp
    return None, False


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
        chunk=64, on_frame=None, write=True, encode_workers=8):
    """Renders every trajectory (one GPU batch per `chunk` poses) and writes the reference's output layout.
    on_frame(trajectory_id, index, rgb uint8 [H,W,3]) — optional — sees each frame as it is handed to the JPEG encoder
    (the array `cam.get_rgba()[:, :, :3]` would be in generate_images.py:428-432).  The array is a VIEW of a pinned ring buffer that
    is overwritten two chunks later: a callback that keeps frames must copy them (as generate_images.py:431 does).

    The host side is a two-deep pipeline: chunk i is packed to uint8 on the GPU and copied into a PINNED buffer on a copy stream
    while chunk i+1 is rendered (Renderer.host_frames), and its JPEGs are encoded by `encode_workers` threads (libjpeg releases the
    GIL) while the GPU works on — in the reference each frame is rendered, read back and encoded strictly in turn
    (generate_images.py:408-436).  write=False skips the encoder (throughput of the render + readback path alone).
    Back-pressure: at most 4 x encode_workers frames wait for (or are in) the encoder — the GPU produces several thousand frames/s, eight
    PIL threads encode 1-1.5 k/s, and every queued frame holds its own 2.4 MB (1024x768) copy: unbounded, a scene of a few thousand
    waypoints held gigabytes per process.  The pool is shut down on every exit path."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(out_dir, exist_ok=True)
    w, h = int(resolution[0]), int(resolution[1])
    sequences, total = [], 0
    pool = ThreadPoolExecutor(max_workers=max(1, int(encode_workers))) if write else None
    pending = []                 # [(handle, [(trajectory id, index, path)])]: copies in flight
    jobs = []                    # encoder futures not yet collected, oldest first (bounded: see the docstring)
    max_jobs = 4 * max(1, int(encode_workers))

    def encode(rgb, path):
        from PIL import Image
        Image.fromarray(rgb).save(path, quality=quality)

    def drain(keep):
        while len(pending) > keep:
            hnd, items = pending.pop(0)
            host = hnd.wait()                                     # [chunk,H,W,4] pinned; valid until the ring comes round again
            for k, (tid, idx, path) in enumerate(items):
                rgb = host[k, :, :, :3]
                if on_frame is not None:
                    on_frame(tid, idx, rgb)
                if pool is not None:
                    while len(jobs) >= max_jobs:
                        jobs.pop(0).result()                      # the encoders are behind: wait for the oldest frame (its error surfaces here)
                    jobs.append(pool.submit(encode, np.ascontiguousarray(rgb), path))

    try:
        # The scene's waypoints are ONE work list cut into chunks, whatever trajectory they belong to: a trajectory holds a dozen or two
        # sampled points (generate_actions.py:586-592), and a batch per trajectory would fill and drain the GPU's frame pipeline every time —
        # the reference's loop (generate_images.py:408-436) renders them one after the other within a scene just the same.
        work = []                    # (trajectory id, index, camera, file path)
        scheduled = set()
        for tr in trajectories:
            tdir = os.path.join(out_dir, "images", f"trajectory_{tr['trajectory_id']}")      # generate_images.py:308,395
            names = [f"{scene_id}_{tr['trajectory_id']}_{i:03d}.jpg" for i in range(len(tr["points"]))]
            done = tdir in scheduled or (os.path.isdir(tdir) and all(os.path.exists(os.path.join(tdir, n)) for n in names))
            if (not done or force) and names:
                scheduled.add(tdir)
                if write:
                    os.makedirs(tdir, exist_ok=True)
                for i, cam in enumerate(cameras_for(tr["points"], resolution)):
                    work.append((tr["trajectory_id"], i, cam, os.path.join(tdir, names[i])))
            total += len(names)
            sequences.append({"scene_id": scene_id, "trajectory_id": tr["trajectory_id"],
                              "instruction_index": tr["instruction_index"], "frame_filenames": names,
                              "trajectory_sampled_points": [{"point_id": p["point"], "position": p["position"],
                                                             "rotation": p["rotation"]} for p in tr["points"]],
                              "sampling_info": {"sampled_points_count": len(names), "generated_images_count": len(names),
                                                "data_source": "sage_gs.sweep"}})
        if work:
            ring = renderer.host_frames((chunk, h, w, 4), depth=2)
            frames = None
            for c0 in range(0, len(work), chunk):
                part = work[c0:c0 + chunk]
                drain(1)                                              # at most one copy in flight beside the batch being rendered
                # [B,H,W,3] on the GPU; ONE pack and ONE device-to-host copy per chunk (B stacked images are one tall image)
                frames = renderer.render_batch([it[2] for it in part], scene,
                                               out=frames if frames is not None and frames.shape[0] >= len(part) else None)
                buf = frames if frames.shape[0] == chunk else torch_pad(frames, chunk)
                pending.append((ring.submit(buf, n=len(part)), [(it[0], it[1], it[3]) for it in part]))
        drain(0)
        for j in jobs:
            j.result()                                            # (an encoder error surfaces here)
    finally:
        if pool is not None:
            pool.shutdown(wait=True, cancel_futures=True)
    meta = {"scene_id": scene_id, "scene_name": scene_id, "total_image_sequences": len(sequences),
            "frames_per_sequence": "variable_based_on_action_sampling", "image_resolution": list(resolution),
            "camera_settings": {"focal_length": CAMERA_FOCAL_LENGTH, "height": CAMERA_HEIGHT},
            "sequences": sequences, "processing_mode": {"type": "gpu_batch", "scene_reuse": True, "single_camera": True}}
    json.dump(meta, open(os.path.join(out_dir, "image_metadata.json"), "w", encoding="utf-8"), ensure_ascii=False, indent=2)
    return total


def torch_pad(frames, chunk):
    """A [chunk,H,W,3] buffer holding `frames` at its head (the ring's buffers have one shape; a trajectory's last batch is short)."""
    import torch
    buf = torch.zeros((chunk,) + tuple(frames.shape[1:]), dtype=frames.dtype, device=frames.device)
    buf[:frames.shape[0]] = frames
    return buf


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--scene", help="one scene: a 3DGS .ply (standard layout) or, with --compressed, a PlayCanvas compressed .ply")
    ap.add_argument("--compressed", action="store_true")
    ap.add_argument("--actions", help="one scene: its action_groundtruth.json")
    ap.add_argument("--scene-id")
    ap.add_argument("--action-root", help="many scenes: <action-root>/<scene_id>/action_groundtruth.json (generate_images.py:194)")
    ap.add_argument("--scene-root", help="many scenes: where <scene_id>.ply / <scene_id>/3dgs[_compressed].ply live")
    ap.add_argument("--instance-id", type=int, default=0, help="this process's share of the scenes (generate_images.py:72-75)")
    ap.add_argument("--total-instances", type=int, default=1)
    ap.add_argument("--out", required=True)
    ap.add_argument("--device", default="auto", help="cuda:N, or auto = cuda:LOCAL_RANK / cuda:(instance_id mod device count)")
    ap.add_argument("--width", type=int, default=CAMERA_RESOLUTION[0])
    ap.add_argument("--height", type=int, default=CAMERA_RESOLUTION[1])
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args(argv)
    if not 0 <= a.instance_id < a.total_instances:
        ap.error(f"instance_id({a.instance_id}) must be in range [0, {a.total_instances})")
    many = a.action_root is not None
    if many == (a.scene is not None) or (many and not a.scene_root) or (not many and not (a.actions and a.scene_id)):
        ap.error("give either --scene/--actions/--scene-id (one scene) or --action-root/--scene-root (many)")
    from . import ply, scenes
    from .renderer import Renderer
    import torch
    dev = a.device
    if dev == "auto":
        dev = f"cuda:{int(os.environ['LOCAL_RANK']) if 'LOCAL_RANK' in os.environ else a.instance_id % max(1, torch.cuda.device_count())}"
    r = Renderer(dev)
    jobs = [(a.scene_id, a.scene, a.compressed, a.actions, a.out)] if not many else []
    if many:
        for sid in scenes_of_instance(a.action_root, a.instance_id, a.total_instances):
            path, comp = find_scene_file(a.scene_root, sid)
            if path is None:
                print(f"[sage_gs.sweep] scene {sid}: no .ply under {a.scene_root}; skipped")       # generate_images.py:157-159
                continue
            jobs.append((sid, path, comp, os.path.join(a.action_root, sid, "action_groundtruth.json"), os.path.join(a.out, sid)))
    total = 0
    for sid, path, comp, actions, out_dir in jobs:
        arrays = (ply.load_compressed_ply if comp else ply.load_ply)(path)
        scene = r.upload(ply.to_gaussians(arrays, dev, scenes.MODEL_TO_WORLD))      # template.usda:120
        n = run(r, scene, load_trajectories(actions), sid, out_dir, (a.width, a.height), a.force)
        scene.free()
        total += n
        print(f"[sage_gs.sweep] instance {a.instance_id + 1}/{a.total_instances}: scene {sid}: {n} frames -> {out_dir}")
    print(f"[sage_gs.sweep] {total} frames of {len(jobs)} scene(s)")


if __name__ == "__main__":
    main()
