"""Device kd build times (rtx_kd_build: levels on the device / whole call) for the parity scenes and the dragon stand-ins."""
import sys
sys.path.insert(0, ".")
from raytracer_amd import scenes
for name in (sys.argv[1:] or ["teapot", "bunny", "dragon1", "dragon2"]):
    best = None
    for _ in range(3):
        sc = scenes.build(name, kd_device=0)
        best = sc.kd_ms_device if best is None else min(best, sc.kd_ms_device)
        st = sc.kd_stats()
        sc.close()
    print("%-8s nodes %8d  device levels + numbering %.1f ms" % (name, st["nodes"], best))
