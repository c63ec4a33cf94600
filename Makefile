# Builds libtennis_hip.so (gfx950 only) in-tree, plus the C oracle pieces.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := tennis_amd/csrc
OUT   := tennis_amd/lib/libtennis_hip.so
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(SRCS:.hip=.o)
HFLAGS := $(EXTRA) --offload-arch=$(ARCH) -O3 -std=c++20 -fPIC -Wall -Wno-unused-function

all: $(OUT)

$(CSRC)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/tennis_hip.h
	$(HIPCC) $(HFLAGS) -c $< -o $@

# the strip kernel pins its own schedule: SLP packing of its scalar f32 arithmetic (v_pk_add_f32 ...) only costs issue slots
$(CSRC)/dense_strip_w56.o $(CSRC)/dense_strip_w28.o $(CSRC)/dense_strip_w128.o $(CSRC)/dense_strip_w64.o: HFLAGS += -fno-slp-vectorize

$(OUT): $(OBJS)
	@mkdir -p $(dir $(OUT))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

clean:
	rm -f $(OBJS) $(OUT)

.PHONY: all clean
