import os, sys
sys.path.insert(0, os.getcwd())
import bench
print(bench.imm_leg())
