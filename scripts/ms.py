"""Print ms_per_step (and the conv roofline) of the last JSON line on stdin (tuning helper)."""
import json, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d.get('roofline') or {}
print(d['ms_per_step'], r.get('achieved'), r.get('share_of_step_ms'))
