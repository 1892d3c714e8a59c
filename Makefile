# Convenience targets; the build itself is __graft_entry__.build() (hipcc --offload-arch=gfx950, in-tree).
PY ?= python3

.PHONY: build test test-gpu bench smoke soak example clean
build:
	$(PY) -c "import __graft_entry__ as g; g.build()"
test: build
	$(PY) -m pytest tests -q -m "not gpu"
test-gpu: build
	$(PY) -m pytest tests -q -m gpu
smoke: build
	$(PY) __graft_entry__.py smoke
bench: build
	$(PY) bench.py
soak: build
	$(PY) tools/soak.py 300
example: build
	gcc -std=c99 -Wall -Wextra -pedantic -Iinclude examples/roundtrip.c -o examples/roundtrip \
	    -Lfusion-cryptography_amd/lib -lfusion_hip -Wl,-rpath,$(CURDIR)/fusion-cryptography_amd/lib
clean:
	rm -f fusion-cryptography_amd/lib/*.so fusion-cryptography_amd/lib/*.o examples/roundtrip
	$(MAKE) -s -C oracle clean || true
