classdef KpOwner < handle
    %KpOwner: lifetime of ONE kp_mex handle (context reference, dictionary, MPC problem, trajectory object, ...).
    %   Ksysid / Kmpc are value classes: they are copied freely and MATLAB calls no destructor on them.  A handle-class
    %   member IS destroyed - when the last copy of the value object that carries it (and the last lift closure that
    %   captured it) is gone - so KsysidHip / KmpcHip keep their device handles inside KpOwner objects and nothing has to be
    %   released by hand.  `parent` keeps the owner of the object a handle points into (a dictionary points into its
    %   context) alive until this one is deleted, which fixes the order of destruction.
    properties ( SetAccess = private )
        value;      % uint64 handle of kp_mex
        command;    % the kp_mex command that releases it
        parent;     % KpOwner that must outlive this one (or [])
    end
    methods
        function obj = KpOwner( value , command , parent )
            obj.value = value;
            obj.command = command;
            if nargin > 2
                obj.parent = parent;
            end
        end
        function release( obj )
            % early, explicit release (delete does the same when the object goes away)
            if ~isempty( obj.value )
                v = obj.value;
                obj.value = [];
                try
                    kp_mex( obj.command , v );
                catch
                    % the MEX file was cleared first: mexAtExit has released everything already
                end
            end
        end
        function delete( obj )
            obj.release();
        end
    end
end
