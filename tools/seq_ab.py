"""benchlib.cfg2.sequence_from_host on its own (4K and 1080p): the two arrangements of the host-fed sequence loop, alternating."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib.cfg2 import link_rates, sequence_from_host     # noqa: E402

if __name__ == "__main__":
    fd = os.dup(1)
    os.dup2(2, 1)
    link = link_rates()
    only = sys.argv[1] if len(sys.argv) > 1 else None           # one arrangement, 4K only, one pass: for a profiler
    if only:
        out = {"4k": sequence_from_host(0, 3840, 2160, 20000, 64, link.get("4k"), alternations=1, only=only)}
    else:
        out = {"4k": sequence_from_host(0, 3840, 2160, 20000, 128, link.get("4k")),
               "1080p": sequence_from_host(0, 1920, 1080, 5000, 256, link.get("1080p"))}
    os.write(fd, (json.dumps(out) + "\n").encode())
