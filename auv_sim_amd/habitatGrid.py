"""Drop-in for habitatGrid.py's `HabitatGrid` (+ the `Habitat` / `HabitatCell` records of habitat.py,
habitatCell.py): the habitat-id layout of the cell grid (habitatGrid.py:14-49, including the way the
row / column habitat counters advance), `within_habitat_env` (:59-65), `inside_habitat` (:68-87),
`distance_from_grid_boundary` (:90-105).  Pure host-side index arithmetic (one int() per query): no
kernel; `habitat_id_grid` is the dense int32 array of the same ids for callers that batch the lookup.
"""
import numpy as np


class Habitat:
    def __init__(self, x, y, id_in, side_length=1, num_of_time_visited=0):
        self.x, self.y, self.id = x, y, id_in
        self.side_length = side_length
        self.num_of_time_visited = num_of_time_visited

    def __repr__(self):
        return ("Habitat: [x=" + str(self.x) + ", y=" + str(self.y) + ", id=" + str(self.id) + ", side length=" +
                str(self.side_length) + ", visited=" + str(self.num_of_time_visited) + "]")

    __str__ = __repr__


class HabitatCell:
    def __init__(self, x, y, habitat_id, side_length=1):
        self.x, self.y = x, y
        self.side_length = side_length
        self.habitat_id = habitat_id

    def __repr__(self):
        return ("Habitat: [id=" + str(self.habitat_id) + ", x=" + str(self.x) + ", y=" + str(self.y) + ", side length=" +
                str(self.side_length) + "]")

    __str__ = __repr__


class HabitatGrid:
    def __init__(self, env_x, env_y, env_size_x, env_size_y, habitat_side_length=10, cell_side_length=1):
        self.env_x, self.env_y = env_x, env_y
        self.env_size_x, self.env_size_y = env_size_x, env_size_y
        self.habitat_side_length = habitat_side_length
        self.cell_side_length = cell_side_length
        per = int(habitat_side_length) // int(cell_side_length)           # cells per habitat side
        n_rows = int(env_size_y) // int(cell_side_length)
        n_cols = int(env_size_x) // int(cell_side_length)
        bands = -(-n_cols // per) if n_cols else 0                        # habitats per band of rows
        # the reference's running counters (:29-44) number the habitats row-major: id = band_row * bands + band_col
        rr, cc = np.arange(n_rows) // per, np.arange(n_cols) // per
        self.habitat_id_grid = (rr[:, None] * bands + cc[None, :]).astype(np.int32)
        xs = [env_x + c * cell_side_length for c in range(n_cols)]
        ys = [env_y + r * cell_side_length for r in range(n_rows)]
        ids = self.habitat_id_grid.tolist()
        self.habitat_cell_grid = [[HabitatCell(xs[c], ys[r], ids[r][c], side_length=cell_side_length)
                                   for c in range(n_cols)] for r in range(n_rows)]
        self.habitat_array = [Habitat(xs[c], ys[r], ids[r][c], habitat_side_length)
                              for r in range(0, n_rows, per) for c in range(0, n_cols, per)]

    def print_habitat_cell_grid(self):
        for ids in self.habitat_id_grid.tolist():
            print(*ids, sep=' ', end=' \n')

    def within_habitat_env(self, auv_pos):
        x, y = auv_pos[0], auv_pos[1]
        return (self.env_x <= x < self.env_x + self.env_size_x) and (self.env_y <= y < self.env_y + self.env_size_y)

    def inside_habitat(self, auv_pos):
        """the HabitatCell under auv_pos, or False past the top / right edge (positions assumed non-negative,
        as in the reference: negative indices wrap like Python lists do)"""
        r, c = int(auv_pos[1] / self.cell_side_length), int(auv_pos[0] / self.cell_side_length)
        if r >= len(self.habitat_cell_grid):
            print("auv is out of the habitat environment bound verticaly")
            return False
        if c >= len(self.habitat_cell_grid[0]):
            print("auv is out of the habitat environment bound horizontally")
            return False
        return self.habitat_cell_grid[r][c]

    def distance_from_grid_boundary(self, auv_pos):
        """[top wall, right wall, bottom wall, left wall]"""
        x, y = auv_pos[0], auv_pos[1]
        return np.array([self.env_y + self.env_size_y - y, self.env_x + self.env_size_x - x, y - self.env_y, x - self.env_x])
