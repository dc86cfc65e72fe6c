import json,sys
d=json.load(open(sys.argv[1]))
print(d["ms_per_step"], json.dumps(d["kernel_families"]))
