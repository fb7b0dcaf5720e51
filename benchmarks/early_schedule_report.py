"""Early-schedule parity report: one guided step at t = T-1 with the eps-consistent synthetic UNet, every record
with its error, peak, criterion and STRICT verdict.  Usage (GPU box): python benchmarks/early_schedule_report.py [mini|cfg256] > out.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import step_checks as sc  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "mini"
for precision in (0, 1):
    if case == "mini":
        recs = sc.check_step("mini", precision, respacing="50", steps=1, t_first=49, head_scale=1.0, eps_consistent=True)
    else:
        recs = sc.check_step("cfg256", precision, respacing="250", steps=1, t_first=249, cutn=16, vit_name="ViT-B/32", head_scale=1.0,
                             scales=(1000.0, 150.0, 50.0), eps_consistent=True)
    for r in recs:
        print(f"{'OK  ' if r['ok'] else 'FAIL'} strict={'yes' if r['ok_strict'] else 'NO '} {r['name']}: abs {r['err_abs']:.3e} "
              f"rel-to-peak {r['err_rel']:.3e} peak {r['ref_max']:.3e} [{r['criterion']}]", flush=True)
