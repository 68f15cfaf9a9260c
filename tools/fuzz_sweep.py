"""One-off wide sweep of the seeded random-configuration test (tests/test_gpu_fuzz.py)
beyond the seeds the suite runs: python tools/fuzz_sweep.py [first last]   (GPU box).
Integration rule = seed mod 3; every second triple of seeds adds a radius-ramp cloud
(and sometimes a transparent core); every third sextuple runs `cut vertical`, the rest the default `cut slant`.  Uses the test's own helper, so the oracle is
involved: this is a test driver, not part of the product."""
import sys, tempfile, pathlib, traceback
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import test_gpu_fuzz as f
bad = 0
first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 560)
for seed in range(first, last):
    d = pathlib.Path(tempfile.mkdtemp(prefix="fz"))
    try:
        f._run(d, seed, integ=seed % 3, ramp=(seed // 3) % 2 == 1, cut="vertical" if (seed // 6) % 3 == 0 else None)
    except Exception as e:
        bad += 1
        print("SEED", seed, "FAILED", str(e)[:600].replace("\n", " | "))
    import shutil; shutil.rmtree(d, ignore_errors=True)
print("sweep done, failures:", bad)
