#!/usr/bin/env python3
"""Run-to-run determinism of the whole training step at small / odd shapes: the same five steps from the same weights
and batch, three times in one process - the loss sequences must be bit-identical (integer statistics accumulators,
fixed-order reductions, no float atomics).  Prints the sequences and whether they agree."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

import torch  # noqa: E402


def main() -> None:
    from ddl import comm
    from ddl.config import TrainConfig
    from ddl.models.densenet import UNIT_NAMES
    from ddl.trainer import SingleStrategy
    from test_gpu_native import _ref_stage
    dev = torch.device("cuda", 0)
    ok = True
    for B, S in ((2, 32), (7, 160), (1, 96), (30, 224)):
        ref = _ref_stage(UNIT_NAMES, dev)
        g = torch.Generator().manual_seed(B * 1000 + S)
        x = torch.randint(0, 256, (B, 3, S, S), dtype=torch.uint8, generator=g).to(dev)
        y = torch.randint(0, 5, (B,), generator=g).to(dev)
        runs = []
        for rep in range(3):
            strat = SingleStrategy(TrainConfig(strategy="single", batch_size=B), comm.CommContext(0, 1, 0, dev, "none"))
            strat.model.import_state_dict(ref.state_dict())
            losses = torch.stack([strat.train_step(x, y, B)[0].clone() for _ in range(6)])
            runs.append(losses.cpu())
            del strat
        same = all(torch.equal(runs[0], r) for r in runs[1:])
        ok &= same
        print(f"B={B} S={S}: {'bit-identical' if same else 'DIFFERENT'}")
        for r in runs:
            print("   ", [f"{float(v):.7f}" for v in r])
    print("deterministic" if ok else "NOT deterministic")


if __name__ == "__main__":
    main()
