cd /root/repo
echo "== product"; python scripts/gpu_wide_tables_concurrent.py | tail -4
echo "== debug library, uploads on the tables' own streams"; WHAMD_USE_DEBUG_LIB=1 WHAMD_UPLOAD_ON_TABLE_STREAM=1 python scripts/gpu_wide_tables_concurrent.py | tail -4
