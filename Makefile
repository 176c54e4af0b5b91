# Same targets as the reference's Makefile (/root/reference/Makefile:1-7): `make image` builds the renderer and
# renders the 1024x768, 16-samples-per-pixel image to out.tga -- here through the MI355X backend.
SHELL := /bin/bash
.PHONY: all rtrace image test clean

all: rtrace

rtrace:
	python3 -c "import __graft_entry__ as g; g.build()"

image: rtrace
	time ./rust-tracer_amd/rtrace --samples-per-pixel=4 --width=1024 --height=768 out.tga

test:
	python3 -m pytest tests -x -q -m "not gpu"

clean:
	$(MAKE) -C rust-tracer_amd/csrc clean
	$(MAKE) -C rust-tracer_amd/csrc/host clean
	$(MAKE) -C oracle clean
