import sys, os
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import trace_conv as tc
tc.run(4, 256, 256, 32, 32, stats=True)
tc.run(8, 256, 256, 32, 32, stats=True)
tc.run(4, 128, 128, 64, 64, stats=True)
tc.run(8, 128, 128, 64, 64, stats=True)
tc.run(4, 64, 64, 128, 128, stats=True)
tc.run(4, 384, 256, 32, 32, stats=True)
