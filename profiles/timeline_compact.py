import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
hits = [i for i, r in enumerate(rows) if "fm2_fwd_bce" in r["Kernel_Name"]]
start = hits[int(sys.argv[2])]
t0 = int(rows[start]["Start_Timestamp"])
ab = {"fm2_fwd_bce": "FWD", "bce_final": "fin", "fg_reduce": "RED", "fg_combine": "cmb", "fg_row_update": "upd", "fg_hist": "h", "fg_tile": "t", "fg_scatter": "s", "fg_bin_wave": "w", "fg_bin_big": "b"}
for r in rows[start:start + int(sys.argv[3])]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = next((v for k, v in ab.items() if k in r["Kernel_Name"]), r["Kernel_Name"][:20])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:6.1f} q{r.get('Queue_Id','?')} {nm}")
