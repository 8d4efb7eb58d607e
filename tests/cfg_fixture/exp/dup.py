_base_ = ["../_base_/misc.py", "../_base_/misc.py"]
