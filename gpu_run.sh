bash tools/r06_final.sh r06_ev2
