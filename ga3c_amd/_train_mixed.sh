mkdir -p checkpoints logs
python "$(dirname "$0")/GA3C_mixed.py" "$@"
