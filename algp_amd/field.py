"""Synthetic field with the data-side interface of the reference's FieldEnv (reference
env.py:15-113, 423-446): X / Y / test_X / test_Y, num_samples, collect_samples, index <-> pose maps.
It exists so the GP path can be driven end to end (BASELINE config 1, a 20 x 20 mixture-of-Gaussians
field) without the reference's networkx path planner, which is outside this package's scope.
"""
import numpy as np

from .utils import generate_gaussian_data


class SyntheticField(object):
    def __init__(self, num_rows=20, num_cols=20, num_test=40, k=5):
        self.num_rows, self.num_cols = num_rows, num_cols
        x, y = generate_gaussian_data(num_rows, num_cols, k=k)          # utils.py:90-108
        x = x.astype(np.float64)
        perm = np.random.permutation(len(x))                            # env.py:60-63
        test_ind, train_ind = perm[:num_test], perm[num_test:]
        self.X, self.Y = x[train_ind], y[train_ind]
        self.test_X, self.test_Y = x[test_ind], y[test_ind]
        self.all_x, self.all_y = np.copy(x), np.copy(y)
        self.map_pose_to_gp_index_matrix = np.full((num_rows, num_cols), None)
        self.gp_index_to_map_pose_array = np.full(len(self.X), None)
        for ind in range(len(self.X)):
            pose = (int(self.X[ind, 0]), int(self.X[ind, 1]))
            self.map_pose_to_gp_index_matrix[pose] = ind
            self.gp_index_to_map_pose_array[ind] = pose

    @property
    def shape(self):
        return (self.num_rows, self.num_cols)

    @property
    def num_samples(self):
        return len(self.X)

    def collect_samples(self, indices, noise_std):
        """Noisy reading, truncated at zero (env.py:108-113)."""
        return max(0, self.Y[indices] + np.random.normal(0, noise_std))

    def gp_index_to_map_pose(self, gp_index):
        return self.gp_index_to_map_pose_array[gp_index]

    def map_pose_to_gp_index(self, map_pose):
        return self.map_pose_to_gp_index_matrix[tuple(map_pose)]
