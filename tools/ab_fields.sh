#!/bin/bash
# Blocks handed over without / with their field arrays: the bench line's value, eval and explicit_fields legs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for a in "--steps 200 --warmup 10" "--steps 20 --warmup 5" "--config c3" "--config c2"; do
  python3 bench.py --no-cpu-baseline $a > /tmp/f.json 2>/dev/null
  python3 - "$a" <<'P'
import json, sys
d = json.load(open('/tmp/f.json'))
print(sys.argv[1], "| value", round(d['value']), d['ms_per_step'], "| explicit", d.get('explicit_fields', {}).get('ms_per_step'),
      "| resident", d.get('resident', {}).get('ms_per_step'), "| eval", round(d['eval']['value']), d['eval'].get('roofline_frac'),
      "resident", round(d['eval']['resident']), d['eval'].get('roofline_frac_resident'), "|", d['config']['field_array'])
P
done
python3 bench.py --no-cpu-baseline --steps 200 --explicit-fields > /tmp/f.json 2>/dev/null
python3 -c "
import json
d = json.load(open('/tmp/f.json'))
print('--explicit-fields | value', round(d['value']), d['ms_per_step'], '| eval', round(d['eval']['value']), d['eval'].get('roofline_frac'), '|', d['config']['field_array'])"
