"""Development aid (GPU box): what one chunk of the bench's staged pass puts on the link, column by column."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["UZ_BENCH_LINK_BYTES"] = "1"
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--steps", "1", "--warmup", "0", "--chunks", "8"]))
