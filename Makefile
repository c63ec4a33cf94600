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

# the strip kernel pins its own schedule: SLP packing of its scalar f32 arithmetic (v_pk_add_f32 ...) only costs issue slots.
# Its units are compiled with -save-temps so that the ISA that IS in the object (not a second compile) can be audited:
# the kernel keeps a window of bottleneck rows in literal accumulator registers behind hipcc's back
# (scripts/audit_strip_isa.py; `make audit`, run by __graft_entry__.build() and tests/test_cpu_build.py).
STRIP_UNITS := dense_strip_w56 dense_strip_w28 dense_strip_w128 dense_strip_w64
STRIP_OBJS  := $(STRIP_UNITS:%=$(CSRC)/%.o)
ISA_DIR     := $(CSRC)/isa
# (grouped target: one recipe makes the object AND its ISA listing)
define STRIP_RULE
$(CSRC)/$(1).o $(ISA_DIR)/$(1).s &: $(CSRC)/$(1).hip $(wildcard $(CSRC)/*.h) include/tennis_hip.h
	@mkdir -p $(ISA_DIR)
	$(HIPCC) $(HFLAGS) -fno-slp-vectorize -save-temps=obj -c $(CSRC)/$(1).hip -o $(CSRC)/$(1).o
	@mv $(CSRC)/$(1)-hip-amdgcn-amd-amdhsa-gfx950.s $(ISA_DIR)/$(1).s
	@rm -f $(CSRC)/$(1)-hip-amdgcn-amd-amdhsa-gfx950.* $(CSRC)/$(1)-host-x86_64-unknown-linux-gnu.* $(CSRC)/$(1).hip-hip-amdgcn-amd-amdhsa.hipfb
endef
$(foreach u,$(STRIP_UNITS) dense_block14 dense_block28,$(eval $(call STRIP_RULE,$(u))))

# dense_block14.hip keeps its activation ring in literal registers v[192:255] and counts its own vmcnt: scripts/audit_block14_isa.py
audit: $(STRIP_UNITS:%=$(ISA_DIR)/%.s) $(ISA_DIR)/dense_block14.s $(ISA_DIR)/dense_block28.s
	python3 scripts/audit_strip_isa.py $(STRIP_UNITS:%=$(ISA_DIR)/%.s)
	python3 scripts/audit_block14_isa.py $(ISA_DIR)/dense_block14.s
	python3 scripts/audit_block14_isa.py $(ISA_DIR)/dense_block28.s

$(OUT): $(OBJS)
	@mkdir -p $(dir $(OUT))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

clean:
	rm -f $(OBJS) $(OUT)
	rm -rf $(ISA_DIR)

.PHONY: all clean audit
