"""prints what a bench.py line says about its attempts:  python tools/show_attempts.py line.json"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value", d["value"], "verified", d["verified"], "|", d.get("value_failed"))
for a in d["attempts"]:
    print(a["attempt"], a["asked_for"], a["layout"], a["outcome"], a["seconds"], "s |", (a["why"] or "")[:120], "|", (a.get("error_lines") or [])[-2:])
print({k: (v if "error" in v else {"value": v.get("value"), "verified": v.get("verified")}) for k, v in (d.get("modes") or {}).items()})
