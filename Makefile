# Convenience targets; everything also works without make (see README.md).
PY ?= python

.PHONY: build oracle test test-gpu bench smoke golden clean

build:            ## hipcc --offload-arch=gfx950 -> mixemt_amd/lib/libmixemt_hip.so (no GPU needed)
	$(PY) -m mixemt_amd.build

oracle:           ## the checker's C restatement -> oracle/_build/liboracle.so
	$(MAKE) -C oracle

test: build oracle  ## CPU suite: oracle vs goldens, host logic, ABI exports, gloo loop
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu: build oracle  ## parity through the C ABI on an MI355X
	$(PY) -m pytest tests -q -m gpu

smoke: build
	$(PY) __graft_entry__.py smoke

bench: build      ## one JSON line: 10^6 reads x 5408 haplogroups, one EM init
	$(PY) bench.py

golden:           ## regenerate tests/golden from the reference (needs /root/reference)
	$(PY) tools/gen_golden.py

clean:
	rm -rf mixemt_amd/lib oracle/_build
