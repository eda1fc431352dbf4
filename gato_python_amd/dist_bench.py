"""bench.py leg for --gpus N > 1 (filled in with the knot-sharded solver, see dist.py)."""


def main(args):
    raise SystemExit("multi-GPU bench: not built yet")
