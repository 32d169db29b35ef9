"""One line per bench JSON: step, dp_rank, dp_rank with the collective stand-in.   python tools/show_standin.py a.json b.json ..."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:  # noqa: BLE001
        print(f, 'unreadable:', e)
        continue
    r = d.get('dp_rank', {})
    s = r.get('standin', {})
    print(f'{f:40s} step {d["ms_per_step"]:.3f} ms | dp_rank {r.get("ms_per_step_dp_rank")} | with stand-in {r.get("ms_per_step_dp_rank_with_standin")} '
          f'({s.get("gbytes_per_s")} GB/s, {s.get("workgroups")} workgroups, {s.get("communicators")} communicator(s)) | ceiling {r.get("scaling_ceiling_with_standin")}')
