% syntax2.gml
%
% bad function syntax

1 { /x x { /y y } apply

