"""ParticleDefaults (reference: src/Operators/core_2D.jl:40-58)."""
from dataclasses import dataclass


@dataclass
class ParticleDefaults:
    lne: float
    c̄_x: float
    c̄_y: float
    x: float = 0.0
    y: float = 0.0

    def as_vector(self):
        return [self.lne, self.c̄_x, self.c̄_y, self.x, self.y]


ParticleDefaults2D = ParticleDefaults
