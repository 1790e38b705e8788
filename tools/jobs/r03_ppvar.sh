for v in "" _NOLOAD _NOREAD _NOBAR _NOLOAD_NOREAD _NOLOAD_NOREAD_NOBAR; do
  if [ -n "$v" ]; then export GROVE_HIP_LIB=$PWD/ppvar/lib$v.so; else unset GROVE_HIP_LIB; fi
  echo "${v:-base}: $(timeout 200 python tools/pp_ablate.py 2>&1 | tail -1)"
done
