import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
for k in d["kernels"]: print(k["name"], k.get("isolated_mean_ms"), k.get("mean_ms"))
print(d.get("match_roofline"))
print(d.get("valu_roofline"))
print(d.get("roofline"))
print(d.get("single_frame_host_to_host_ms"), d.get("track_frame_host_to_host_ms"))
