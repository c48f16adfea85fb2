"""The detailed record of a bench.py run for the profile tools: either the side file bench.py writes (SK_BENCH_DETAILS: {"headline": {...}, "c2": ...})
or a one-line record of rounds 1-4.  Returns the headline's record with `roofline` = the trailing SYRK's object (round 5 calls it roofline_syrk)."""
import json


def load(path):
    text = open(path).read().strip()
    try:
        raw = json.loads(text)
    except ValueError:
        raw = json.loads(text.splitlines()[-1])
    line = raw.get("headline", raw)
    if "roofline_syrk" in line:
        line = dict(line, roofline=line["roofline_syrk"])
    return line
