"""summary of tools/r06_planner_sweep.sh: the planner's modulus against every forced one, per grid (to_cc + from_cc of all spin groups)"""
import sys, os, re, json, collections
best = collections.OrderedDict(); chosen = {}
for line in open("gpurun_out/r06_planner_sweep/sweep_%s.txt" % os.environ.get("TAG", "a")):
    if line.startswith("=="): print(line.strip()[:200]); continue
    m = re.match(r"(\w+) g=(\d+) (\{.*\}) \| (.*)", line)
    if not m:
        print("  (no result)", line[:80].strip()); continue
    cfg, g, d, ch = m.group(1), int(m.group(2)), json.loads(m.group(3)), m.group(4)
    ms = d["ms"]; t = sum(v for k, v in ms.items() if k.startswith(("to_cc", "from_cc_nc")))
    best.setdefault(cfg, []).append((g, round(t, 3)))
    if g == 0:
        mm = re.search(r"g=(\d+)", ch); chosen[cfg] = mm.group(1) if mm else "?"
for cfg, rows in best.items():
    dflt = [t for g, t in rows if g == 0][0]; forced = [(g, t) for g, t in rows if g != 0]
    if forced:
        bg, bt = min(forced, key=lambda x: x[1]); print("%s: planner chose g=%s: %.3f ms; best forced g=%d: %.3f ms; choice/best = %.3f; all %s" % (cfg, chosen.get(cfg), dflt, bg, bt, dflt/bt, rows))
    else: print("%s: planner chose g=%s: %.3f ms; no other modulus realises ducc0's N_cc" % (cfg, chosen.get(cfg), dflt))
