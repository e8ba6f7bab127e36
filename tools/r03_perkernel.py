"""print the per-kernel table of a bench.py JSON line: python tools/r03_perkernel.py gpurun_out/x.json"""
import json, sys
for p in sys.argv[1:]:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    print(p, d["value"], "frames/s", d["ms_per_step"], "ms", d["stages"])
    for k in d["roofline"]["per_kernel"]:
        print("   %-50s x%d  %7.1f us  %6.1f TF  frac %.4f" % (k["kernel"], k["launches_per_step"], k["avg_us"], k["tflops"], k["frac"]))
