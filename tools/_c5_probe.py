import sys, json
sys.path.insert(0,'.')
import bench
from algp_amd import _hip
print(json.dumps(bench.extra_c5(_hip, 0, 4)))
