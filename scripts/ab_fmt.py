import json, sys
d = json.loads(sys.stdin.read().strip().split("\n")[-1])
print("%-34s %9.1f frames/s  %s" % (sys.argv[1], d["value"], {k[:12]: v for k, v in d["roofline"]["families_ms_per_step"].items()}))
