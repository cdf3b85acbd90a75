timeout 600 python -m pytest tests -m gpu -q --tb=short -k "checkpointing or rollout or rccl or multistep" 2>&1 | tail -30
