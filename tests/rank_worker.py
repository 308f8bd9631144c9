"""One rank process of a multi-process test: `python rank_worker.py <module> <function> <rank> <json args>` imports
`<module>` from tests/ and calls `<function>(rank, *args)` -- what torch.multiprocessing.start_processes did with a
spawned interpreter, but started by tests/launcher.py instead of by the (GPU-initialised) pytest process."""
import faulthandler
import importlib
import json
import os
import sys


def main():
    faulthandler.enable()
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    module, function, rank, args = sys.argv[1], sys.argv[2], int(sys.argv[3]), json.loads(sys.argv[4])
    getattr(importlib.import_module(module), function)(rank, *args)


if __name__ == "__main__":
    main()
