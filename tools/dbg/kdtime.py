#!/usr/bin/env python3
"""Host vs device SAH kd build times.  usage: kdtime.py [scene ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import scenes, hip
hip.device_count()
for name in sys.argv[1:] or ["teapot", "bunny", "dragon1"]:
    t0 = time.time(); a = scenes.build(name); t1 = time.time()
    b = scenes.build(name, kd_device=0); t2 = time.time()
    c = scenes.build(name, kd_device=0); t3 = time.time()
    sa, sb = a.kd_stats(), b.kd_stats()
    print("%s: %d nodes %d refs | scene build with host kd %.2f s, with device kd %.2f s (first), %.2f s (second); device levels %.1f ms | same tree: %s"
          % (name, sa["nodes"], sa["refs"], t1 - t0, t2 - t1, t3 - t2, c.kd_ms_device, sa == sb), flush=True)
    a.close(); b.close(); c.close()
