# Convenience targets; the authoritative entry points are __graft_entry__.build(), pytest and bench.py.
.PHONY: build test test-gpu bench clean
build:
	python __graft_entry__.py
test: build
	python -m pytest tests -q -m "not gpu"
test-gpu: build
	python -m pytest tests -q -m gpu
bench: build
	python bench.py
clean:
	rm -f hardware-efficient-mua-compression_amd/libmuahuff.so oracle/libmh_oracle.so examples/abi_roundtrip
