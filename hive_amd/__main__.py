"""``python -m hive_amd`` = the reference's ``python -m hive`` (/root/reference/hive/__main__.py:17-20)."""
from hive_amd.pipeline import main

if __name__ == '__main__':
    main()
