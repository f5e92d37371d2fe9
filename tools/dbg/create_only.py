import sys, os
sys.path.insert(0, os.getcwd())
from raytracer_amd import hip, scenes
sc = scenes.build("teapot")
print("built", flush=True)
ds = hip.DeviceScene(sc.desc, 0)
print("created", flush=True)
ds.close()
