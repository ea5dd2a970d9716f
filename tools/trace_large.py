import sys, os
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import trace_conv as tc
tc.run(4, 32, 32, 256, 256, stats=True)
tc.run(4, 32, 32, 256, 256, stats=True, prologue=True)
tc.run(4, 64, 32, 256, 256, stats=True)
tc.run(4, 32, 32, 128, 128, stats=True)
