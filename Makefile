# Convenience targets; __graft_entry__.build() is what the driver calls.
PYTHON ?= python3

build:
	$(PYTHON) -c "import __graft_entry__ as g; g.build()"

test:            # CPU-side suite (oracle pins, ABI, host logic, gloo sharding)
	$(PYTHON) -m pytest tests -x -q -m "not gpu"

test-gpu:        # parity through the C-ABI; needs an MI355X
	$(PYTHON) -m pytest tests -x -q -m gpu

bench:
	$(PYTHON) bench.py

sanitize:        # host C under ASan + UBSan on the CPU suite, in a scratch copy
	bash tools/sanitize_cpu_tests.sh

clean:
	$(MAKE) -C canvas_amd/csrc clean
	$(MAKE) -C canvas_amd/pyext clean
	$(MAKE) -C oracle clean

.PHONY: build test test-gpu bench sanitize clean
