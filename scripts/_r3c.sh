bash scripts/ab_libs.sh 2>&1
