#!/bin/bash
# the bench.py lines behind the table in profiles/README.md (one GPU, 1 GiB per call unless said otherwise)
run() { echo "## bench.py $*"; timeout -k 10 400 python bench.py "$@" 2>&1 | tail -1; }
run --steps 10 --warmup 2
run --steps 5 --level 2 --no-cpu --no-extra
run --steps 5 --level 0 --gen random --no-cpu --no-extra
run --steps 3 --level 0 --gen random --mib 8192 --no-cpu --no-extra
run --steps 5 --level 1 --gen random --no-cpu --no-extra
run --steps 5 --level 2 --gen random --no-cpu --no-extra
run --steps 5 --level 1 --gen log --format gzip --no-cpu --no-extra
run --steps 5 --level 2 --gen log --no-cpu --no-extra
run --steps 5 --level 1 --gen mix --no-cpu --no-extra
run --steps 5 --level 3 --gen mix --no-cpu --no-extra
run --steps 5 --level 4 --gen mix --no-cpu --no-extra
run --steps 5 --level 5 --gen mix --no-cpu --no-extra
run --steps 5 --level 6 --gen mix --no-cpu --no-extra
run --steps 5 --level 1 --warm 4096 --no-cpu --no-extra
run --steps 5 --level 1 --warm 32768 --no-cpu --no-extra
run --steps 5 --level 6 --no-cpu --no-extra
run --steps 3 --level 1 --mib 8192 --no-cpu --no-extra
run --steps 10 --level 1 --inflight 2 --no-cpu --no-extra
