import json,sys
j=json.loads(open(sys.argv[1]).readline())
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j.get("cpu_baseline",{}).get("value"))
x=j.get("extras",{})
def show(k,v,ind=""):
    if isinstance(v,dict):
        keys={kk:vv for kk,vv in v.items() if kk in ("value","unit","tick_us","ms_per_step","metric")}
        print(ind+k, keys)
        for kk,vv in v.items():
            if isinstance(vv,dict) and kk not in ("roofline","cpu_baseline","config"): show(kk,vv,ind+"  ")
    else: print(ind+k, v if not isinstance(v,list) else len(v))
for k,v in x.items(): show(k,v)
