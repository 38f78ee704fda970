"""Headless ``DataRepository``: the list layout of the reference's ``src/models/data_repository.py:11-29``
(``pc_*_list[0]`` = original cloud, ``[1:]`` = HEM levels) without Qt."""


class DataRepository:
    def __init__(self):
        self.pc_gaussian_list_first = []
        self.pc_gaussian_list_second = []
        self.pc_open3d_list_first = []
        self.pc_open3d_list_second = []
        self.current_index = 0


class UIStateRepository:
    def __init__(self):
        import numpy as np
        self.transformation_matrix = np.eye(4)
