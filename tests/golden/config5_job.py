"""One oracle run of tests/golden/make_config5_golden.py, kept on disk: `python config5_job.py SEED MEMBER OUTDIR` writes
OUTDIR/job_SEED_MEMBER.npz (skipped when it exists).  make_config5_golden.py --assemble OUTDIR builds the fixtures from
such files -- the runs take 20 minutes to hours each, so they are made resumable one by one."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_config5_golden as g  # noqa: E402

seed, member, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
path = os.path.join(outdir, "job_%d_%d.npz" % (seed, member))
if not os.path.exists(path):
    r = g.one_run((seed, member))
    np.savez_compressed(path + ".tmp.npz", **{k: np.asarray(v) for k, v in r.items()})
    os.replace(path + ".tmp.npz", path)
print(path)
