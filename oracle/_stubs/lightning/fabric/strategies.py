class DeepSpeedStrategy:  # only used in isinstance checks (lit_llama/utils.py:51)
    pass


class FSDPStrategy:  # only used in isinstance checks (lit_llama/utils.py:63)
    pass
