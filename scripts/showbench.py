import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"]), d["rank0"], d["config"]["optimal_cost_checksum"], d.get("end_to_end",{}).get("value"))
    elif "whatshap_amd" in l or "Error" in l: print(l.strip())
