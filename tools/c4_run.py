import sys, json; sys.path.insert(0, '/root/repo')
from orb_slam2_map_amd import workloads
r = workloads.c4(); r.pop("per_keyframe")
print(json.dumps(r))
