#!/usr/bin/env python3
"""The stress scene with real-asset triangle statistics beside the headline scene (round 5, VERDICT r4 task 6): for each of
atrium (262 k uniform triangles), stadium (272 k) and stadium2m (2.2 M) one bench.py run in sets of frames -- ms per frame, Grays/s, the stages,
node steps / triangle tests per ray and the lanes live per wave step from the counting re-walk -- and the same run with the shadow cache
off (what the cache is worth on each).   usage: tools/stress_scene.py [steps] > profiles/r05/stress_scene.txt"""
import json
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40


def run(scene, cache):
    env = dict(os.environ)
    if not cache:
        env["RT_DEBUG_OPTIONS"] = "shadow_cache_res=0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--scene", scene, "--steps", str(steps), "--warmup", "20", "--no-live-pmc", "--hbm-frames", "0",
           "--cpu-seconds", "0", "--no-strong"] + ([] if cache else ["--no-frame-by-frame"])
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-2000:])
        raise SystemExit(1)
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


print("# tools/stress_scene.py %d: bench.py --scene X --steps %d --warmup 20 (sets of frames), and RT_DEBUG_OPTIONS=shadow_cache_res=0 for the last column" % (steps, steps))
for scene in ("atrium", "stadium", "stadium2m"):
    d = run(scene, True)
    off = run(scene, False)
    print("\n%s: %s" % (scene, d["config"]["workload"]))
    print("  frame %.3f ms in sets = %.0f Mrays/s (%.2f M rays per frame); frame by frame %.3f ms; BVH build %.2f ms warm; shadow cache off: %.3f ms in sets (cache worth %.1f %%)"
          % (d["ms_per_step"], d["value"], d["rays_per_frame"] / 1e6, d.get("frame_by_frame", {}).get("ms_per_frame", float("nan")), d["bvh_rebuild_ms"],
             off["ms_per_step"], 100.0 * (off["ms_per_step"] / d["ms_per_step"] - 1.0)))
    for name, st in d["stages"].items():
        if not isinstance(st, dict) or "walk" not in st:
            continue
        w = st["walk"]
        lanes_node = (w["nodes_global"] + w["nodes_lds"]) / max(64.0 * w["wave_node_steps"], 1.0)
        lanes_tri = w["tris"] / max(64.0 * w["wave_tri_steps"], 1.0)
        print("  %-10s %.3f ms  %6.0f Mrays/s  rays %9d  node steps / ray %.1f (from LDS %.1f)  triangle tests / ray %.2f  longest walk %d  lanes per node step %.2f, per triangle step %.2f"
              % (name, st["avg_ms"], st["Mrays_per_s"], st["rays"], st["nodes_global_per_ray"] + st["nodes_lds_per_ray"], st["nodes_lds_per_ray"], st["tris_per_ray"],
                 st["longest_walk_steps"], lanes_node, lanes_tri))
    sys.stdout.flush()
