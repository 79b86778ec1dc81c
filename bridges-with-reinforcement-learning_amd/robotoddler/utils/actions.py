"""generate_actions / filter_actions (robotoddler/utils/actions.py:7-82 of the reference) for the single-environment
API.  (The vectorised path does both on the device: k_enumerate, k_raster.)"""
import numpy as np
import torch

from assembly_gym.envs.gym_env import Action, AssemblyGym


def generate_actions(gym: AssemblyGym, x_discr_ground, offset_values=None, max_angle_rad=2 * np.pi + 0.1,
                     max_blocks_per_face=1, include_frozen=False, x_block_offset=None):
    if offset_values is None:
        offset_values = [0.]
    if include_frozen:
        raise NotImplementedError
    for shape_index, shape in enumerate(gym.shapes):
        for face in shape.target_faces_2d:
            for offset_x in x_discr_ground:
                yield Action(-1, 0, shape_index, face, offset_x, offset_y=0.)
            # (arccos is at most pi: the default 2 pi + 0.1 excludes nothing and the face frames need not be built)
            check_angle = max_angle_rad is not None and max_angle_rad < np.pi
            for target_block, block in enumerate(gym.assembly_env.blocks):
                for target_face in block.receiving_faces_2d:
                    if check_angle and np.arccos(np.clip(block.get_face_frame_2d(target_face).normal[2], -1.0, 1.0)) > max_angle_rad:
                        continue
                    if max_blocks_per_face and len(gym.block_graph.get((target_block, target_face), ())) >= max_blocks_per_face:
                        continue
                    for offset_x in offset_values:
                        yield Action(target_block, target_face, shape_index, face, offset_x, offset_y=0.)


def filter_actions(gym_env, available_actions, action_features, block_features, obstacle_features, xlim, ylim):
    """Keep actions that stay in bounds and overlap neither the state nor the obstacle raster (actions.py:71-82 of the
    reference).  Same three tests per action; the two raster overlaps of ALL actions are two reductions and one copy
    to the host (the reference's loop reads two device scalars back per action), and collision_on_action finds the
    blocks get_action_features created a moment ago in the gym's candidate cache."""
    n = len(available_actions)
    if n == 0:
        return [], action_features[:0]
    in_bounds = torch.tensor([not gym_env.collision_on_action(a, xlim, ylim) for a in available_actions], dtype=torch.bool)
    feats = action_features.reshape(n, -1)
    free = ((feats * block_features.reshape(1, -1)).sum(dim=1) == 0) & ((feats * obstacle_features.reshape(1, -1)).sum(dim=1) == 0)
    mask = in_bounds & free.cpu()
    kept = [a for a, m in zip(available_actions, mask.tolist()) if m]
    return kept, action_features[mask.to(action_features.device)]
