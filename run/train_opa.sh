#!/bin/bash
# OPA LoRA-SFT stage — same variables / flags as the reference's run/train_opa.sh (one process per MI355X, RCCL).
set -e
export GPUS_PER_NODE=${GPUS_PER_NODE:-8}
export DATA_DIR=${DATA_DIR:-"./base_datasets/opa_training_data-7B"}
export IMAGE_DIR=${IMAGE_DIR:-"none"}
export MODEL_DIR=${MODEL_DIR:-"./base_models/llava-v1.5-7b"}
export OUTPUT_DIR=${OUTPUT_DIR:-"./output/llava7b_opa_model"}
export PYTHONPATH="$PWD:$PYTHONPATH"
export HSA_ENABLE_IPC_MODE_LEGACY=0

PERDEVICE_BS=${PERDEVICE_BS:-4}
GRADIENT_ACC=${GRADIENT_ACC:-8}
EPOCH=${EPOCH:-2}
ENTROPY_LOSS=${ENTROPY_LOSS:-"False"}
ENTROPY_MASK_RAIO=${ENTROPY_MASK_RAIO:-0.8}
ENTROPY_MASK_METHOD=${ENTROPY_MASK_METHOD:-"random"}
ENTROPY_LOSS_COEF=${ENTROPY_LOSS_COEF:-0.01}
ENTROPY_DECAY_COEF=${ENTROPY_DECAY_COEF:-1.0}
LORA_RANK=${LORA_RANK:-256}
LORA_ALPHA=${LORA_ALPHA:-512}

python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node=$GPUS_PER_NODE \
    ./opadpo/opa_train_custom.py \
    --cfg 'configs/llava/llava_opa.yaml' \
    --base_model $MODEL_DIR --output_dir $OUTPUT_DIR --image_folder $IMAGE_DIR --data_dir $DATA_DIR \
    --per_device_train_batch_size $PERDEVICE_BS --per_device_eval_batch_size $PERDEVICE_BS \
    --gradient_accumulation_steps $GRADIENT_ACC --tf32 --bf16 --use_flash_attention --save_steps 40 --eval_steps 10 \
    --mm_vision_select_layer -2 --mm_projector_type "mlp2x_gelu" \
    --full_tune False --tune_mm_mlp_adapter True --tune_base_model True --tune_vision_tower True \
    --lora_tune True --lora_rank $LORA_RANK --lora_alpha $LORA_ALPHA --lora_drop 0.0 --num_train_epochs $EPOCH \
    --entropy_loss $ENTROPY_LOSS --entropy_mask_ratio $ENTROPY_MASK_RAIO --entropy_mask_method $ENTROPY_MASK_METHOD \
    --entropy_loss_coef $ENTROPY_LOSS_COEF --entropy_decay_coef $ENTROPY_DECAY_COEF "$@"
