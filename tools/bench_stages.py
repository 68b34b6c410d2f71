import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f))
    x=d.get("bf16x3", d)
    print(f, d["value"], "| x3", x["value"], x["ms_per_step"])
    for s in x["stages"]: print("  ", s["stage"], round(s["avg_ms"],3), round(s["frac"],3), s.get("clock_mhz"), s.get("frac_at_clock"))
