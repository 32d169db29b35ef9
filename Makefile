# Drop-in for the training part of the reference Makefile (variables and `train` target of
# /root/reference/Makefile:36-59,83-85).  `make build` compiles the gfx950 kernels first.

DATA_DIR ?= ./data

ifdef CLUSTER_SPEC
	CLUSTER_PARAM1 = --cluster-spec=${CLUSTER_SPEC}
else
	CLUSTER_PARAM1 :=
endif
ifdef JOB_TYPE
	CLUSTER_PARAM2 = --job-name=${JOB_TYPE}
else
	CLUSTER_PARAM2 :=
endif
ifdef TASK_INDEX
	CLUSTER_PARAM3 = --task-index=${TASK_INDEX}
else
	CLUSTER_PARAM3 :=
endif
CLUSTER_PARAMS ?= ${CLUSTER_PARAM1} ${CLUSTER_PARAM2} ${CLUSTER_PARAM3}

# Default training parameters
MODEL ?= msdn
RUNID ?= ''
STEPS ?= 10000000
BATCHSIZE ?= 32
DATASET ?= nyu
SUM_FREQ ?= 100
CKPT_FREQ ?= 900
CKPT_DIR ?= checkpoints
TIMEOUT ?= 4200
# data-parallel replicas on this node (replaces PS_NODES / WORKERS of `make distributed`)
GPUS ?= 1
WORKERS ?= 8

# Preprocessing sizes (the reference's Makefile:46-59)
WIDTH ?= 640
HEIGHT ?= 480
DHEIGHT ?= 55
DWIDTH ?= $(shell echo $$(( ${DHEIGHT} * ${WIDTH} / ${HEIGHT} )))
FORCE ?=
START ?= 0
LIMIT ?=

# `make convert nyu make3d1` / `make preprocess nyu`: dataset names as extra goals, like the reference (Makefile:68-78)
ifeq (preprocess,$(firstword $(MAKECMDGOALS)))
    ifneq (,$(wordlist 2,$(words $(MAKECMDGOALS)),$(MAKECMDGOALS)))
        DATASET := $(wordlist 2,$(words $(MAKECMDGOALS)),$(MAKECMDGOALS))
        $(eval $(DATASET):;@:)
    endif
endif
ifeq (convert,$(firstword $(MAKECMDGOALS)))
    ifneq (,$(wordlist 2,$(words $(MAKECMDGOALS)),$(MAKECMDGOALS)))
        DATASET := $(wordlist 2,$(words $(MAKECMDGOALS)),$(MAKECMDGOALS))
        $(eval $(DATASET):;@:)
    endif
endif

SCRIPT_PARAMETERS := --ckptdir=${CKPT_DIR} --datadir=${DATA_DIR} --model=${MODEL} --id=${RUNID} \
					 --steps=${STEPS} --batchsize=${BATCHSIZE} --ckptfreq=${CKPT_FREQ} \
					 --sumfreq=${SUM_FREQ} --timeout=${TIMEOUT} ${CLUSTER_PARAMS}

ifeq (${GPUS},1)
SCRIPT := python3 -O -m ann3depth_amd.ann3depth
else
SCRIPT := python3 -O -m torch.distributed.run --nnodes=1 --nproc-per-node ${GPUS} --master-addr 127.0.0.1 \
		  -m ann3depth_amd.ann3depth
endif

.PHONY: train
train: ${DATA_DIR}
	${SCRIPT} ${SCRIPT_PARAMETERS} ${DATASET}

# raw downloads -> PNG pairs of one size (the reference's `make preprocess`, Makefile:118-121), without h5py / scipy.misc;
# NYU only (tools/data_preprocessor.py)
.PHONY: preprocess
preprocess: ${DATA_DIR}
	DATA_DIR=${DATA_DIR} WIDTH=${WIDTH} HEIGHT=${HEIGHT} DHEIGHT=${DHEIGHT} DWIDTH=${DWIDTH} FORCE=${FORCE} START=${START} LIMIT=${LIMIT} python3 tools/data_preprocessor.py $(DATASET)

# convert preprocessed PNG pairs to TFRecord shards (the reference's `make convert`, Makefile:123-126), TF-free
.PHONY: convert
convert: ${DATA_DIR}
	DATA_DIR=${DATA_DIR} python3 tools/data_tf_converter.py $(DATASET) --del_raw

# the reference's `make distributed` submitted parameter servers and workers to a Sun Grid Engine (Makefile:87-96); here
# the same number of workers are synchronous RCCL replicas on this node
.PHONY: distributed
distributed:
	$(MAKE) train GPUS=${WORKERS}

# tensorboard on the checkpoint directory (event files are written by ann3depth_amd/summary.py), Makefile:129-131
.PHONY: tb
tb:
	tensorboard --logdir=${CKPT_DIR}

.PHONY: help
help:
	${SCRIPT} --help

.PHONY: build
build:
	$(MAKE) -C ann3depth_amd/csrc

${DATA_DIR}:
	mkdir -p ${DATA_DIR}
