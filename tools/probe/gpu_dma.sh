tools/ab_run.sh dma base stg10 stg20 stg40 2>&1 | grep rep
