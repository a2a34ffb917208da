# product launch time of the literal configs[1] shape vs launcher block target and triangle staging
for R in 256 1024 4096; do
  for st in 0 1; do
    for b in 640 1280 2560 5120; do
      DRT_DENSE_STAGE=$st DRT_DENSE_BLOCKS=$b scratch/literal_lab $R
    done
  done
done
for st in 0 1; do DRT_DENSE_STAGE=$st scratch/literal_lab 65536; done
