python "$(dirname "$0")/GA3C.py" PLAY_MODE=True "$@"
