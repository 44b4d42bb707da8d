import glob
import gzip
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pagemgr_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "pagemgr_*.json.gz")))


def load(path):
    with gzip.open(path, "rb") as f:
        return json.loads(f.read().decode())
