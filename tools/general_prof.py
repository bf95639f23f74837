"""Developer tool: one equality-form solve with the blocked column search, for rocprofv3 --kernel-trace --stats.  usage: general_prof.py [mg ng reps general_block]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomilp_amd import lp
mg = int(sys.argv[1]) if len(sys.argv) > 1 else 500
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 700
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
gb = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(5)
A0 = np.zeros((2 * mg, ng + mg)); A0[:mg, :ng] = rng.standard_normal((mg, ng)); A0[mg:, :ng] = rng.standard_normal((mg, ng)); A0[mg:, ng:] = np.eye(mg)
for i in range(reps):
    print(len(lp.find_independent_device(A0, general_block=gb)))
