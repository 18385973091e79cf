"""get_logger of the reference's util/logger.py (a named stream logger; nothing on the hot path)."""
import logging


def get_logger():
    logger = logging.getLogger(name='DPS')
    if not logger.handlers:
        logger.setLevel(logging.INFO)
        handler = logging.StreamHandler()
        handler.setFormatter(logging.Formatter("%(asctime)s [%(name)s] >> %(message)s"))
        logger.addHandler(handler)
    return logger
