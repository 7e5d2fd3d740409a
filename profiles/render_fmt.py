import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(d["scene"], d["width"], d["strata_per_pixel"], round(d["gpu_samples_per_s"]/1e6,1), round(d["gpu_Mrays_per_s"]))
