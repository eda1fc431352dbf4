# Convenience targets; the authoritative entry points are __graft_entry__.py (build, smoke), bench.py and pytest.
.PHONY: build test test-gpu bench smoke clean
build:
	python -c "import __graft_entry__ as g; g.build()"
test: build
	python -m pytest tests -x -q -m "not gpu"
test-gpu: build
	python -m pytest tests -x -q -m gpu
smoke: build
	python -c "import __graft_entry__ as g; g.smoke()"
bench: build
	python bench.py
clean:
	$(MAKE) -C gato_python_amd/csrc clean
	$(MAKE) -C oracle clean
	rm -rf bindings/pybind11/build examples/solve_pendulum
	$(MAKE) -C bindings/fastseq clean
