"""Trainer -> inference-worker weight hand-off when both share ONE MI355X (BASELINE.json configs[2]'s
colocated layout), over HIP IPC: two processes, the request carries the bucket handles, the worker
copies device-to-device into its own weights.  Prints one JSON line per parameter set.

    python scripts/wsync_colocated_bench.py [--sets 0p5b,7b] [--iters 5]
"""

import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


from pipelinerl_amd.weight_sync_probe import colocated_probe  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="0p5b,7b")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    for which in args.sets.split(","):
        for rehome, direct in ((False, False), (True, False), (False, True), (True, True)):
            print(json.dumps(colocated_probe(which, args.iters, rehome, direct=direct)), flush=True)


if __name__ == "__main__":
    main()
