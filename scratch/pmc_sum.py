import csv, glob, collections, sys
for d in sys.argv[1:]:
    f = glob.glob(f"gpurun_out/{d}/**/*_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv_mfma_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"{k:36s} {sum(v)/len(v):16.0f}  n={len(v)}")
