# Convenience targets (the build itself is dsabeamformer_amd/build.py: hipcc for gfx950, in-tree libdsabf.so + beam,
# junkdb, beam_replicas; oracle/Makefile builds the CPU oracle for the tests).  The reference's own `make` / `make debug`
# split (makefile:13-26) is a run-time choice here: `beam -p ... -d ... -s ... -o data.py` is the DEBUG run, `beam -j N`
# / `beam -k ring` the observation mode.
PY ?= python

all:
	$(PY) -c "import __graft_entry__ as g; g.build()"

test: all
	$(PY) -m pytest tests -x -q -m "not gpu"

gputest: all
	$(PY) -m pytest tests -x -q -m gpu

bench: all
	$(PY) bench.py

debug: all            # the reference's `make debug && bin/beam`: DEBUG run on the linear configuration, writes data.py
	dsabeamformer_amd/beam -p tests/golden/config/linear_positions.txt -d tests/golden/config/linear_directions.txt \
		-s tests/golden/config/linear_source_directions_1024.txt -o data.py

clean:
	rm -rf dsabeamformer_amd/build dsabeamformer_amd/libdsabf.so dsabeamformer_amd/beam dsabeamformer_amd/junkdb \
		dsabeamformer_amd/beam_replicas oracle/liborc.so tests/support/libfakerccl.so

.PHONY: all test gputest bench debug clean
